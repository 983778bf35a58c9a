#!/bin/bash
# GPU box: EVERY detector kernel of the f16x4 engine on a 32-page batch (four CRAFT groups of 8 pages), per-kernel average per launch; knobs as arguments
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export TTR_PREC=f16x4
rm -rf /tmp/c32; rocprofv3 --kernel-trace --stats -d /tmp/c32 -o s --output-format csv -- python3 $R/tools/prof_pages.py 32 4 28 detector_only=1 "$@" > /tmp/c32.log 2>&1
f=$(ls /tmp/c32/*/*kernel_stats.csv /tmp/c32/*kernel_stats.csv 2>/dev/null | tail -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"kernel time per 32-page pass {tot/4e6:.3f} ms")
for r in rows[:24]:
    print(f"{float(r['TotalDurationNs'])/4e6:8.3f} ms/pass {int(r['Calls'])/4:5.1f} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
