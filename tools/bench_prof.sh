#!/bin/bash
# GPU box: rocprofv3 kernel stats of the default bench command -> gpurun_out/bench_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/bp; rocprofv3 --kernel-trace --stats -d /tmp/bp -o s --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > /tmp/bp.log 2>&1
tail -1 /tmp/bp.log | cut -c1-300
f=$(ls /tmp/bp/*/*kernel_stats.csv /tmp/bp/*kernel_stats.csv 2>/dev/null | tail -1)
mkdir -p $R/gpurun_out; cp $f $R/gpurun_out/bench_stats.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['Percentage']):5.1f}% {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:80]}")
PY
