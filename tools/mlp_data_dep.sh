#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/mdd; rocprofv3 --kernel-trace -d /tmp/mdd -o s --output-format csv -- python3 $R/tools/mlp_data_dep.py ${1:-20} > /dev/null 2>&1
f=$(ls /tmp/mdd/*/*kernel_trace.csv /tmp/mdd/*kernel_trace.csv 2>/dev/null | tail -1)
python3 - "$f" ${1:-20} <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mlp_fused" in r["Kernel_Name"]]
n = int(sys.argv[2])
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
for k, name in enumerate(("random", "same rows", "zero rows")):
    seg = d[k * n:(k + 1) * n]
    print(f"{name:10s} launches {len(seg)}  median {sorted(seg)[len(seg)//2]:.1f} us  min {min(seg):.1f}  max {max(seg):.1f}")
PY
