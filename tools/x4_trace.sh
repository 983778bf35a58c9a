#!/bin/bash
# GPU box: kernel traces of the f16x4 engine: detector alone (7 blank pages = one CRAFT group) and recogniser alone (1280 crops)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export TTR_PREC=${TTR_PREC:-f16x4}
rm -rf /tmp/tp; rocprofv3 --kernel-trace -d /tmp/tp -o t --output-format csv -- python3 $R/tools/prof_pages.py ${PAGES:-7} 3 0 "$@" > /tmp/tp.log 2>&1
tail -1 /tmp/tp.log
f=$(ls /tmp/tp/*/*kernel_trace.csv /tmp/tp/*kernel_trace.csv 2>/dev/null | tail -1)
python3 $R/tools/trace_seq.py $f resize_pad 0
rm -rf /tmp/tq; rocprofv3 --kernel-trace -d /tmp/tq -o t --output-format csv -- python3 $R/tools/prof_parseq.py 1280 3 "$@" > /tmp/tq.log 2>&1
tail -1 /tmp/tq.log
f=$(ls /tmp/tq/*/*kernel_trace.csv /tmp/tq/*kernel_trace.csv 2>/dev/null | tail -1)
python3 $R/tools/trace_seq.py $f patchify 1
