#!/bin/bash
# GPU box: the round's profile set -> gpurun_out/ (copy what is to be judged into profiles/)
#   1. rocprofv3 --pmc MFMA counters over the recogniser alone (1280 crops)   2. kernel stats of the default bench command
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
rm -rf /tmp/pm; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d /tmp/pm -o p -- python3 $R/tools/prof_parseq.py 1280 3 > /tmp/pm.log 2>&1
cc=$(ls /tmp/pm/*/*counter_collection.csv /tmp/pm/*counter_collection.csv 2>/dev/null | tail -1); kt=$(ls /tmp/pm/*/*kernel_trace.csv /tmp/pm/*kernel_trace.csv 2>/dev/null | tail -1)
cp $cc $O/pmc_mfma_parseq_counter_collection.csv
python3 $R/tools/pmc_mfma_to_json.py $cc "$kt" $O/pmc_mfma_parseq.json "build $(cd $R && cat .build_hash 2>/dev/null)" | tail -8
rm -rf /tmp/bp; rocprofv3 --kernel-trace --stats -d /tmp/bp -o s --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/bp.log 2>&1
tail -1 /tmp/bp.log | cut -c1-200
cp $(ls /tmp/bp/*/*kernel_stats.csv /tmp/bp/*kernel_stats.csv 2>/dev/null | tail -1) $O/bench_kernel_stats.csv
head -8 $O/bench_kernel_stats.csv | cut -c1-160
