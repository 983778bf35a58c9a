cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export TTR_PREC=f16x4
rm -rf /tmp/tq; rocprofv3 --kernel-trace -d /tmp/tq -o t --output-format csv -- python3 $R/tools/prof_parseq.py 1280 3 > /tmp/tq.log 2>&1
f=$(ls /tmp/tq/*/*kernel_trace.csv /tmp/tq/*kernel_trace.csv 2>/dev/null | tail -1)
python3 $R/tools/trace_seq.py $f patchify 1 > $R/gpurun_out/parseq_seq_full.txt
python3 - $R/gpurun_out/parseq_seq_full.txt <<'PY'
import sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
lines=open(sys.argv[1]).read().split("\n")
print(lines[0])
for l in lines[1:]:
    p = l.split()
    if len(p) < 4: continue
    us, cnt, name = float(p[0]), int(p[2][1:]), " ".join(p[3:])[:70]
    agg[name][0] += cnt; agg[name][1] += us
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{us:9.1f} us {c:5d} x  {k}")
PY
