#!/bin/bash
# PMC passes on the selection kernels alone (tools/select_bench.py), separate passes, kernel-trace only.
#   tools/select_pmc.sh <tag> [LPF_CFG]  ->  gpurun_out/<tag>_select_pmc.txt
TAG=${1:-r04}
export LPF_CFG=${2:-collab}
export LPF_REPS=3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_selpmc
rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/select_bench.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/select_bench.py > $O/write.log 2>&1
python3 $R/tools/pmc_traffic.py $O/fetch $O/write $O/traffic.json ${3:-unknown} > $R/gpurun_out/${TAG}_select_pmc_${LPF_CFG}.txt 2>&1
for K in select3_run_kernel select3_plan_kernel; do
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/sq1 -- python3 $R/tools/select_bench.py > $O/sq1.log 2>&1
python3 $R/tools/pmc_summary.py $O/sq1 $K >> $R/gpurun_out/${TAG}_select_pmc_${LPF_CFG}.txt 2>&1
done
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/tools/select_bench.py > $O/sq2.log 2>&1
python3 $R/tools/pmc_summary.py $O/sq2 select3_run_kernel >> $R/gpurun_out/${TAG}_select_pmc_${LPF_CFG}.txt 2>&1
cp $O/traffic.json $R/gpurun_out/${TAG}_select_pmc_${LPF_CFG}.json
rm -rf $O
cat $R/gpurun_out/${TAG}_select_pmc_${LPF_CFG}.txt
