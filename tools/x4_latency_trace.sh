#!/bin/bash
# GPU box: kernel trace of single-page calls (one synthetic 1024x768 page, 40 words) of the f16x4 engine: launch-order listing of the last call
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export TTR_PREC=${TTR_PREC:-f16x4}
rm -rf /tmp/tl; rocprofv3 --kernel-trace -d /tmp/tl -o t --output-format csv -- python3 $R/tools/prof_pages.py 1 6 40 "$@" > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log
f=$(ls /tmp/tl/*/*kernel_trace.csv /tmp/tl/*kernel_trace.csv 2>/dev/null | tail -1)
python3 $R/tools/trace_seq.py $f resize_pad 1 > /tmp/tl_seq.txt; head -1 /tmp/tl_seq.txt
python3 - /tmp/tl_seq.txt <<'PY'
import sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for l in open(sys.argv[1]).read().split("\n")[1:]:
    p = l.split()
    if len(p) < 4: continue
    us, cnt, name = float(p[0]), int(p[2][1:]), " ".join(p[3:])[:60]
    agg[name][0] += cnt; agg[name][1] += us
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{us:9.1f} us {c:5d} x  {k}")
PY
