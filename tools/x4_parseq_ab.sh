#!/bin/bash
# GPU box: PARSeq-only kernel totals of the f16x4 engine at 1280 crops for a list of tuning settings:  tools/x4_parseq_ab.sh "k=v k=v" "k=v" ...
# (per setting: the launch count and kernel time of one pass, then the totals per kernel)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export TTR_PREC=f16x4
for cfg in "$@"; do
  rm -rf /tmp/tq; rocprofv3 --kernel-trace -d /tmp/tq -o t --output-format csv -- python3 $R/tools/prof_parseq.py ${CROPS:-1280} 3 $cfg > /tmp/tq.log 2>&1
  f=$(ls /tmp/tq/*/*kernel_trace.csv /tmp/tq/*kernel_trace.csv 2>/dev/null | tail -1)
  python3 $R/tools/trace_seq.py $f patchify 1 > /tmp/tq_seq.txt 2>/dev/null
  echo "=== $cfg: $(head -1 /tmp/tq_seq.txt)"
  python3 - /tmp/tq_seq.txt <<'PY'
import sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for l in open(sys.argv[1]).read().split("\n")[1:]:
    p = l.split()
    if len(p) < 4: continue
    us, cnt, name = float(p[0]), int(p[2][1:]), " ".join(p[3:])[:60]
    agg[name][0] += cnt; agg[name][1] += us
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(__import__("os").environ.get("TOP", "8"))]:
    print(f"{us:9.1f} us {c:5d} x  {k}")
PY
done
