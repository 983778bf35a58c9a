#!/bin/bash
# GPU box: SQ stall buckets / instruction mix of the detector's kernels (one 8-page CRAFT group, f16x4 engine) -> gpurun_out/${TAG:-r04}_pmc_stall_craft_*
# (TAG=r03 reproduces round 3's file names; extra arguments are tuning keys for tools/prof_pages.py, e.g. c3_c128_waves=8)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
export TTR_PREC=f16x4
rm -rf /tmp/pc1; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pc1 -o p -- python3 $R/tools/prof_pages.py 8 3 0 "$@" > /tmp/pc1.log 2>&1
cp $(ls /tmp/pc1/*/*counter_collection.csv /tmp/pc1/*counter_collection.csv 2>/dev/null | tail -1) $O/${TAG:-r04}_pmc_stall_craft_sq_counter_collection.csv || tail -5 /tmp/pc1.log
rm -rf /tmp/pc3; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d /tmp/pc3 -o p -- python3 $R/tools/prof_pages.py 8 3 0 "$@" > /tmp/pc3.log 2>&1
cp $(ls /tmp/pc3/*/*counter_collection.csv /tmp/pc3/*counter_collection.csv 2>/dev/null | tail -1) $O/${TAG:-r04}_pmc_stall_craft_inst_counter_collection.csv || tail -5 /tmp/pc3.log
python3 $R/tools/pmc_stall_summary.py $O/${TAG:-r04}_pmc_stall_craft_sq_counter_collection.csv $O/${TAG:-r04}_pmc_stall_craft_inst_counter_collection.csv
