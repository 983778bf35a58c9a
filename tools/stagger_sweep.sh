#!/bin/bash
# GPU box: mlp_fused / qkv_attn launch durations under rocprofv3 for a few stagger settings (groups << 16 | microseconds)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for st in 0 $((2<<16|20)) $((2<<16|40)) $((3<<16|25)) $((4<<16|20)) $((2<<16|60)); do
  rm -rf /tmp/stg; rocprofv3 --kernel-trace --stats -d /tmp/stg -o s --output-format csv -- python3 $R/tools/prof_parseq.py 1280 3 mlp_stagger=$st > /dev/null 2>&1
  f=$(ls /tmp/stg/*/*kernel_stats.csv /tmp/stg/*kernel_stats.csv 2>/dev/null | tail -1)
  echo "stagger=$st ($((st>>16)) groups x $((st&65535)) us)"; grep -E "mlp_fused|qkv_attn" $f | cut -d, -f1-4 | cut -c1-150
done
