#!/bin/bash
# PMC passes on the pair-stage kernels (GPU box, repo root).  Separate passes, kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r02
rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
pass() { # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 $R/tools/fused_variants.py > $O/$n.log 2>&1
  python3 $R/tools/pmc_summary.py $O/$n select_run pair_fused_kernel tail_chain > $O/$n.txt 2>&1
  rm -rf $O/$n
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
cd $R; for f in gpurun_out/pmc_r02/*.txt; do echo "== $f"; head -60 $f; done
