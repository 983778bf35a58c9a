#!/bin/bash
# GPU box: where the waves of the recogniser's kernels spend their cycles (SQ stall buckets, LDS array activity, L2 hits) -> gpurun_out/r03_pmc_stall_*
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
export TTR_PREC=f16x4
rocprofv3 -L > $O/pmc_list_full.txt 2>&1
rm -rf /tmp/ps1; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/ps1 -o p -- python3 $R/tools/prof_parseq.py ${CROPS:-1280} 2 $@ > /tmp/ps1.log 2>&1
cp $(ls /tmp/ps1/*/*counter_collection.csv /tmp/ps1/*counter_collection.csv 2>/dev/null | tail -1) $O/r03_pmc_stall_sq_counter_collection.csv || tail -5 /tmp/ps1.log
rm -rf /tmp/ps2; rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/ps2 -o p -- python3 $R/tools/prof_parseq.py ${CROPS:-1280} 2 $@ > /tmp/ps2.log 2>&1
cp $(ls /tmp/ps2/*/*counter_collection.csv /tmp/ps2/*counter_collection.csv 2>/dev/null | tail -1) $O/r03_pmc_stall_tcc_counter_collection.csv || tail -5 /tmp/ps2.log
rm -rf /tmp/ps3; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d /tmp/ps3 -o p -- python3 $R/tools/prof_parseq.py ${CROPS:-1280} 2 $@ > /tmp/ps3.log 2>&1
cp $(ls /tmp/ps3/*/*counter_collection.csv /tmp/ps3/*counter_collection.csv 2>/dev/null | tail -1) $O/r03_pmc_stall_inst_counter_collection.csv || tail -5 /tmp/ps3.log
python3 $R/tools/pmc_stall_summary.py $O/r03_pmc_stall_sq_counter_collection.csv $O/r03_pmc_stall_tcc_counter_collection.csv $O/r03_pmc_stall_inst_counter_collection.csv
