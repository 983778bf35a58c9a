#!/bin/bash
# GPU box: bench.py under two settings of one tuning knob, alternating, on ONE box.   tools/ab_bench.sh knob a b [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 ${4:-2}); do for v in $2 $3; do
  timeout 400 python $R/bench.py --no-cpu-baseline --latency-iters 0 --tune $1=$v 2>&1 | tail -1 > /tmp/ab.json
  python3 -c "import json; d=json.load(open('/tmp/ab.json')); print('$1=$v', round(d['value'],1), d['stage_ms_last_pass'])"
done; done
