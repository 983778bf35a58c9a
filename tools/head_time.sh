#!/bin/bash
# GPU box: durations of the detector's kernels on a 16-page detector-only batch (rocprofv3 --stats), optional knobs key=value
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ht; rocprofv3 --kernel-trace --stats -d /tmp/ht -o s --output-format csv -- python3 $R/tools/prof_pages.py 16 4 0 "$@" > /dev/null 2>&1
f=$(ls /tmp/ht/*/*kernel_stats.csv /tmp/ht/*kernel_stats.csv 2>/dev/null | tail -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per 16-page pass {tot/4e6:.2f} ms")
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs'])/4e6:8.3f} ms/pass {int(r['Calls'])/4:5.1f} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:80]}")
PY
