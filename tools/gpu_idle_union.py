"""Reads a rocprofv3 kernel-trace CSV of a bench.py run (kernels of several streams may overlap) and prints, for the steady-state part of the run, the wall
time, the time at least one kernel was running (union of the intervals), the idle time and the largest idle gaps with the kernels around them.
   python tools/gpu_idle_union.py <kernel_trace.csv> [skip_fraction=0.4]"""
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
t0, t1 = rows[0][0], rows[-1][1]
lo, hi = t0 + (t1 - t0) * skip, t0 + (t1 - t0) * 0.95
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
busy, gaps, cur_s, cur_e, last_name = 0, [], sel[0][0], sel[0][1], sel[0][2]
for s, e, n in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name[:40], n[:40]))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = n
busy += cur_e - cur_s
wall = sel[-1][1] - sel[0][0]
ksum = sum(e - s for s, e, _ in sel)
print(f"{len(sel)} kernels in {wall / 1e6:.1f} ms: at least one running {busy / 1e6:.1f} ms ({100 * busy / wall:.1f} %), idle {(wall - busy) / 1e6:.2f} ms ({100 * (wall - busy) / wall:.2f} %), "
      f"sum of durations {ksum / 1e6:.1f} ms (overlap factor {ksum / busy:.2f})")
big = sorted(gaps, reverse=True)[:12]
print("largest idle gaps (us): " + "; ".join(f"{g / 1e3:.0f} [{a} -> {b}]" for g, a, b in big))
print(f"gaps > 100 us: {sum(1 for g, _, _ in gaps if g > 1e5)} totalling {sum(g for g, _, _ in gaps if g > 1e5) / 1e6:.2f} ms; gaps <= 100 us: {sum(g for g, _, _ in gaps if g <= 1e5) / 1e6:.2f} ms")
