#!/bin/bash
# GPU box: the benchmark's timed region (no CPU leg, no extras) once per tuning setting, settings interleaved:  tools/ab_tune.sh "k=v" "k=v k2=v" ...
# prints value (pages/s), the dominant kernel's average launch, the last pass's stage times
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  args=""; for kv in $cfg; do args="$args --tune $kv"; done
  python3 $R/bench.py --no-cpu-baseline --no-extras ${STEPS:+--steps $STEPS} $BENCH_ARGS $args 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg:', round(d['value'], 1), 'pages/s; dominant kernel', round(d['roofline']['avg_launch_us'], 1), 'us; stages', d['stage_ms_last_pass'])"
done
