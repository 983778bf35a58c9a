#!/bin/bash
# GPU box: the two HBM-traffic passes over a 16-page detector-only batch (separate --pmc runs, kernel trace only) -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pc_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pc_$c -o p -- python3 $R/tools/prof_pages.py 16 3 0 > /tmp/pc_$c.log 2>&1
  cp $(ls /tmp/pc_$c/*/*counter_collection.csv /tmp/pc_$c/*counter_collection.csv 2>/dev/null | tail -1) $O/pmc_craft_b16_$(echo $c | tr A-Z a-z | sed s/_size//)_counter_collection.csv
done
python3 $R/tools/pmc_to_json.py $O/pmc_craft_b16_fetch_counter_collection.csv $O/pmc_craft_b16_write_counter_collection.csv 16 3 $O/pmc_craft_b16.json | tail -3
python3 $R/tools/pmc_layers_to_json.py $O/pmc_craft_b16_fetch_counter_collection.csv $O/pmc_craft_b16_write_counter_collection.csv 16 $O/pmc_craft_b16.json "$(cat $R/.build_hash 2>/dev/null)" | tail -3
ls -la $O | tail -5
