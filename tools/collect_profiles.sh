#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r02 <commit>   ->   gpurun_out/<tag>_*  (copy the summaries into profiles/ afterwards)
set -u
TAG=${1:-r05}
COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
mkdir -p $OUT
# 1. the headline bench line (default flags)
python3 bench.py > $OUT/${TAG}_bench.log 2>&1; grep '^{"metric"' $OUT/${TAG}_bench.log | tail -1 > $OUT/${TAG}_final_bench.json
# 2. kernel trace + stats of the same command, strictly serial (one stream) so durations are per kernel
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py --launch eager --streams 1 --rows off --weights random --no-cpu-baseline > $OUT/${TAG}_under_rocprof.log 2>&1
grep '^{"metric"' $OUT/${TAG}_under_rocprof.log | tail -1 > $OUT/${TAG}_final_bench_under_rocprof.json
cp $(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_final_kernel_stats.csv
# 2b. the same with the other form of the attention (pair-major rows / unit-major records), serial
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_rows -- python3 bench.py --launch eager --streams 1 --rows on --no-cpu-baseline --weights random --no-bf16 > /dev/null 2>&1
cp $(find $OUT/${TAG}_stats_rows -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_final_kernel_stats_rows_form.csv
rm -rf $OUT/${TAG}_stats_rows
# 3. HBM traffic counters, separate passes (no trace domains besides kernel-trace); the encoder runs in these too
for ROWS in on off; do
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch/$ROWS -- python3 bench.py --launch eager --streams 1 --rows $ROWS --weights random --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline --no-kernel-timing --no-bf16 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write/$ROWS -- python3 bench.py --launch eager --streams 1 --rows $ROWS --weights random --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline --no-kernel-timing --no-bf16 > /dev/null 2>&1
done
python3 tools/pmc_traffic.py $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_pmc_traffic_collab.json $COMMIT collab
# 4. matrix-core utilisation of the shipped MFMA kernels
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/${TAG}_mfma -- python3 bench.py --launch eager --streams 1 --rows on --weights random --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline --no-kernel-timing --no-bf16 > /dev/null 2>&1
python3 tools/pmc_summary.py $OUT/${TAG}_mfma pair_fused_kernel pair_flip_kernel pair_rows_kernel tail_chain_kernel dense_chain_kernel gcn_fused_kernel gemm_f32 > $OUT/${TAG}_pmc_mfma_util.txt 2>&1
echo "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); collected at commit $COMMIT" >> $OUT/${TAG}_pmc_mfma_util.txt
# 5. timeline of the pipelined run
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_tl -- python3 bench.py --launch plan --weights random --no-cpu-baseline --no-kernel-timing --no-bf16 --repeats 1 --steps 40 > /dev/null 2>&1
python3 tools/timeline.py $(find $OUT/${TAG}_tl -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_timeline_pipelined.txt 2>&1
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_tl $OUT/${TAG}_mfma
ls -la $OUT | grep ${TAG}_
