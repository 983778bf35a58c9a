#!/bin/bash
# GPU box: one synchronous single-page call of the f16x4 engine, every launch in order with its duration and the idle gap in front of it
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export TTR_PREC=${TTR_PREC:-f16x4}
rm -rf /tmp/tl; rocprofv3 --kernel-trace -d /tmp/tl -o t --output-format csv -- python3 $R/tools/prof_pages.py 1 6 40 "$@" > /tmp/tl.log 2>&1
f=$(ls /tmp/tl/*/*kernel_trace.csv /tmp/tl/*kernel_trace.csv 2>/dev/null | tail -1)
python3 - $f <<'PY'
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))) for r in csv.DictReader(open(sys.argv[1]))))
a = [i for i, r in enumerate(rows) if "resize_pad" in r[2]][-1]
seq = rows[a:]
print(f"{len(seq)} launches, kernel time {sum(e - s for s, e, *_ in seq) / 1e6:.2f} ms, wall {(seq[-1][1] - seq[0][0]) / 1e6:.2f} ms")
i_ccl = next(i for i, r in enumerate(seq) if "ccl_init" in r[2]); i_pf = next(i for i, r in enumerate(seq) if "patchify" in r[2])
print(f"detector {sum(e - s for s, e, *_ in seq[:i_ccl]) / 1e3:.0f} us, boxes {sum(e - s for s, e, *_ in seq[i_ccl:i_pf]) / 1e3:.0f} us, recogniser {sum(e - s for s, e, *_ in seq[i_pf:]) / 1e3:.0f} us")
import os
if os.environ.get("SUMMARY"): sys.exit(0)
prev = seq[0][0]
for s, e, n, g, w in seq:
    n = n.replace("void ttr::", "").replace("(ttr::ConvParams)", "").replace("_ZN3ttr12_GLOBAL__N_1", "")[:64]
    print(f"{(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f}  grid {g:>7} x {w:<4} {n}")
    prev = e
PY
