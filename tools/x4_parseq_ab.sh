#!/bin/bash
# GPU box: PARSeq-only kernel totals of the f16x4 engine at 1280 crops for a list of tuning settings:  tools/x4_parseq_ab.sh "k=v k=v" "k=v" ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export TTR_PREC=f16x4
for cfg in "$@"; do
  rm -rf /tmp/tq; rocprofv3 --kernel-trace -d /tmp/tq -o t --output-format csv -- python3 $R/tools/prof_parseq.py ${CROPS:-1280} 3 $cfg > /tmp/tq.log 2>&1
  f=$(ls /tmp/tq/*/*kernel_trace.csv /tmp/tq/*kernel_trace.csv 2>/dev/null | tail -1)
  echo "=== $cfg: $(python3 $R/tools/trace_seq.py $f patchify 1 2>/dev/null | head -1)"
  python3 $R/tools/trace_seq.py $f patchify 1 2>/dev/null | sed -n 6,16p
done
