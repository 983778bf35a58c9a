#!/bin/bash
# GPU box: encoder kernel durations on random crops vs flat grey crops (same binary, same launches): the share of MFMA time that is power
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in "" const; do
  rm -rf /tmp/cvr; rocprofv3 --kernel-trace --stats -d /tmp/cvr -o s --output-format csv -- python3 $R/tools/prof_parseq.py 1280 40 $mode > /dev/null 2>&1
  f=$(ls /tmp/cvr/*/*kernel_stats.csv /tmp/cvr/*kernel_stats.csv 2>/dev/null | tail -1)
  echo "crops: ${mode:-random}"; grep -E "mlp_fused|qkv_attn|gemm2_kernel<128, 128" $f | cut -d, -f1-4 | cut -c1-160
done
