#!/bin/bash
# GPU box: per-kernel time of ONE recogniser pass at 1280 crops (rocprofv3 --stats over 10 passes; knobs as arguments)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s --output-format csv -- python3 $R/tools/prof_parseq.py ${CROPS:-1280} 10 "$@" > /dev/null 2>&1
f=$(ls /tmp/pp/*/*kernel_stats.csv /tmp/pp/*kernel_stats.csv 2>/dev/null | tail -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per pass {tot/1e7:.2f} ms, {sum(int(r['Calls']) for r in rows)/10:.0f} launches")
for r in rows[:18]:
    print(f"{float(r['TotalDurationNs'])/1e7:8.3f} ms/pass {int(r['Calls'])/10:6.1f} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:90]}")
PY
