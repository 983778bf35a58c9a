"""Reads a rocprofv3 kernel trace CSV and prints, for the busiest 60 % of the run, the GPU busy fraction and the largest idle gaps.
python tools/gpu_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
lo, hi = t0 + (t1 - t0) * 0.45, t0 + (t1 - t0) * 0.75
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
busy = sum(e - s for s, e, _ in sel)
gaps = []
for a, b in zip(sel, sel[1:]):
    g = b[0] - a[1]
    if g > 0: gaps.append((g, a[2][:50], b[2][:50]))
span = sel[-1][1] - sel[0][0]
print(f"window {span/1e6:.1f} ms, kernels {len(sel)}, busy {busy/span:.4f}, idle {(span-busy)/1e6:.2f} ms; gaps > 20 us: {sum(g for g,_,_ in gaps if g > 20000)/1e6:.2f} ms")
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"  {g/1e3:8.1f} us  after {a}  before {b}")
