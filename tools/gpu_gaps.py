"""Reads a rocprofv3 kernel-trace CSV of a bench.py run and prints, per 32-page step (from one resize launch to the next), the wall
time, the sum of kernel durations, the idle time and the largest gaps.   python tools/gpu_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
starts = [i for i, r in enumerate(rows) if "resize_pad" in r[2]]
print(f"{len(rows)} kernels, {len(starts)} resize launches")
for a, b in zip(starts, starts[1:]):
    if b - a < 300:
        continue                      # single-page latency runs
    span = rows[b][0] - rows[a][0]
    busy = sum(e - s for s, e, _ in rows[a:b])
    gaps = sorted(((rows[i + 1][0] - rows[i][1], rows[i][2][:30], rows[i + 1][2][:30]) for i in range(a, b)), reverse=True)
    idle = sum(g for g, _, _ in gaps if g > 0)
    print(f"step {span / 1e6:6.2f} ms  kernels {busy / 1e6:6.2f} ms  idle {idle / 1e6:5.2f} ms   largest gaps (us): " +
          "; ".join(f"{g / 1e3:.0f} {x} -> {y}" for g, x, y in gaps[:2]))
