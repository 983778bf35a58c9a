#!/bin/bash
# GPU box: detector kernels on a 32-page batch (two CRAFT groups), per-kernel time per pass, knobs as arguments
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/c32; rocprofv3 --kernel-trace --stats -d /tmp/c32 -o s --output-format csv -- python3 $R/tools/prof_pages.py 32 4 0 "$@" > /dev/null 2>&1
f=$(ls /tmp/c32/*/*kernel_stats.csv /tmp/c32/*kernel_stats.csv 2>/dev/null | tail -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
conv = [r for r in rows if any(k in r["Name"] for k in ("conv3p", "gemm2_kernel", "conv3s", "upsample", "maxpool"))]
print(f"detector conv/pool/upsample kernel time per 32-page pass {sum(float(r['TotalDurationNs']) for r in conv)/4e6:.3f} ms")
for r in conv[:14]:
    print(f"{float(r['TotalDurationNs'])/4e6:8.3f} ms/pass {int(r['Calls'])/4:5.1f} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:70]}")
PY
