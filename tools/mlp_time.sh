#!/bin/bash
# GPU box: per-kernel durations of the recogniser at 1280 crops (rocprofv3 --stats), optional tuning knobs as arguments
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/mt; rocprofv3 --kernel-trace --stats -d /tmp/mt -o s --output-format csv -- python3 $R/tools/prof_parseq.py 1280 10 "$@" > /dev/null 2>&1
f=$(ls /tmp/mt/*/*kernel_stats.csv /tmp/mt/*kernel_stats.csv 2>/dev/null | tail -1)
grep -E "mlp_fused|qkv_attn" $f | cut -d, -f1-4 | cut -c1-160
