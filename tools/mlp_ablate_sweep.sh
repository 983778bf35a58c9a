#!/bin/bash
# GPU box (library built with -DMLP_ABLATE_BUILDS): mlp_fused duration with parts removed, production code otherwise
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for a in ${ABLS:-0 2 12 14}; do
  rm -rf /tmp/mab; rocprofv3 --kernel-trace --stats -d /tmp/mab -o s --output-format csv -- python3 $R/tools/prof_parseq.py 1280 10 mlp_ablate=$a > /dev/null 2>&1
  f=$(ls /tmp/mab/*/*kernel_stats.csv /tmp/mab/*kernel_stats.csv 2>/dev/null | tail -1)
  echo "ablate=$a"; grep -E "mlp_fused" $f | cut -d, -f1-4 | cut -c1-160
done
