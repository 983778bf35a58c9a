"""Launch-order listing of the LAST pass in a rocprofv3 kernel-trace CSV: every kernel from the last launch whose name contains
<marker> (default: resize_pad for a page pass, patchify for a recogniser pass), consecutive launches of one kernel folded.
   python tools/trace_seq.py <kernel_trace.csv> [marker] [fold=1]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
marker = sys.argv[2] if len(sys.argv) > 2 else "resize_pad"
fold = int(sys.argv[3]) if len(sys.argv) > 3 else 1
starts = [i for i, r in enumerate(rows) if marker in r[2]]
a = starts[-1]
seq = rows[a:]
tot = sum(e - s for s, e, _ in seq)
print(f"{len(seq)} launches, kernel time {tot/1e6:.2f} ms, wall {(seq[-1][1]-seq[0][0])/1e6:.2f} ms")
out = []
for s, e, n in seq:
    n = n.replace("void ttr::", "").replace("(ttr::ConvParams)", "")[:60]
    if fold and out and out[-1][0] == n:
        out[-1][1] += 1; out[-1][2] += e - s
    else:
        out.append([n, 1, e - s])
for n, c, t in out:
    print(f"{t/1e3:9.1f} us  x{c:<4d} {n}")
