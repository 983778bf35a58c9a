#!/bin/bash
# GPU box: profile set of a round (RT=r05 by default) of the default (f16x4) engine -> gpurun_out/${RT}_*  (copy what is to be judged into profiles/)
#   1. kernel stats of the default bench command           2. MFMA counters over the recogniser alone (1280 crops)
#   3. HBM FETCH / WRITE passes over one 8-page CRAFT group (+ the per-layer table)      4. MFMA counters over that group
#   5. launch-order traces: the recogniser at 1280 crops, one synchronous single-page call
# PMC passes run on their own (--pmc with --kernel-trace only), the program itself behind `--`.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
RT=${RT:-r05}
O=$R/gpurun_out; mkdir -p $O
export TTR_PREC=f16x4
B="build $(cat $R/.build_hash 2>/dev/null)"
rm -rf /tmp/bp; rocprofv3 --kernel-trace --stats -d /tmp/bp -o s --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /tmp/bp.log 2>&1
tail -1 /tmp/bp.log | cut -c1-200
cp $(ls /tmp/bp/*/*kernel_stats.csv /tmp/bp/*kernel_stats.csv 2>/dev/null | tail -1) $O/${RT}_bench_kernel_stats_v1.csv
head -12 $O/${RT}_bench_kernel_stats_v1.csv | cut -c1-170
rm -rf /tmp/pm; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d /tmp/pm -o p -- python3 $R/tools/prof_parseq.py 1280 3 > /tmp/pm.log 2>&1
cc=$(ls /tmp/pm/*/*counter_collection.csv /tmp/pm/*counter_collection.csv 2>/dev/null | tail -1); kt=$(ls /tmp/pm/*/*kernel_trace.csv /tmp/pm/*kernel_trace.csv 2>/dev/null | tail -1)
cp $cc $O/${RT}_pmc_mfma_parseq_x4_counter_collection.csv
python3 $R/tools/pmc_mfma_to_json.py $cc "$kt" $O/${RT}_pmc_mfma_parseq_x4.json "$B; f16x4 engine, 1280 crops" | tail -14
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pc_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pc_$c -o p -- python3 $R/tools/prof_pages.py 8 3 0 > /tmp/pc_$c.log 2>&1
  cp $(ls /tmp/pc_$c/*/*counter_collection.csv /tmp/pc_$c/*counter_collection.csv 2>/dev/null | tail -1) $O/${RT}_pmc_craft_x4_$(echo $c | tr A-Z a-z | sed s/_size//)_counter_collection.csv
done
python3 $R/tools/pmc_to_json.py $O/${RT}_pmc_craft_x4_fetch_counter_collection.csv $O/${RT}_pmc_craft_x4_write_counter_collection.csv 8 3 $O/${RT}_pmc_craft_x4.json | tail -2
python3 $R/tools/pmc_layers_to_json.py $O/${RT}_pmc_craft_x4_fetch_counter_collection.csv $O/${RT}_pmc_craft_x4_write_counter_collection.csv 8 $O/${RT}_pmc_craft_x4.json "$B" | tail -40
rm -rf /tmp/pmc_c; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d /tmp/pmc_c -o p -- python3 $R/tools/prof_pages.py 8 3 0 > /tmp/pmc_c.log 2>&1
cc=$(ls /tmp/pmc_c/*/*counter_collection.csv /tmp/pmc_c/*counter_collection.csv 2>/dev/null | tail -1); kt=$(ls /tmp/pmc_c/*/*kernel_trace.csv /tmp/pmc_c/*kernel_trace.csv 2>/dev/null | tail -1)
cp $cc $O/${RT}_pmc_mfma_craft_x4_counter_collection.csv
python3 $R/tools/pmc_mfma_to_json.py $cc "$kt" $O/${RT}_pmc_mfma_craft_x4.json "$B; f16x4 engine, one 8-page CRAFT group" | tail -8
bash $R/tools/x4_parseq_breakdown.sh > $O/${RT}_parseq_1280_kernel_trace.txt 2>&1; head -14 $O/${RT}_parseq_1280_kernel_trace.txt | cut -c1-120
bash $R/tools/x4_latency_trace.sh > $O/${RT}_single_page_kernel_trace.txt 2>&1; head -14 $O/${RT}_single_page_kernel_trace.txt | cut -c1-120
