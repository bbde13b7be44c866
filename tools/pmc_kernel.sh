#!/bin/bash
# PMC passes on ONE kernel of the collab-like step (GPU box, repo root): where do its cycles go?
#   tools/pmc_kernel.sh <kernel-name fragment> [LPF_CFG]     Separate passes, kernel-trace only.
K=${1:-pair_flip_kernel}
export LPF_CFG=${2:-collab}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$K
rm -rf $O; mkdir -p $O
pass() { # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 $R/tools/fused_variants.py > $O/$n.log 2>&1
  python3 $R/tools/pmc_summary.py $O/$n $K > $O/$n.txt 2>&1
  rm -rf $O/$n
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM
pass sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_VALU
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $R; for f in gpurun_out/pmc_$K/*.txt; do echo "== $f"; head -14 $f; done
