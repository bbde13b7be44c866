#!/bin/bash
# lpf_gemm_tn_f32's partial kernel alone (rocprofv3 kernel trace of tools/gemm_tn_bench.py): the shipped build, the matrix
# work alone (-DLPF_TN_NOLOAD) and the loads alone (-DLPF_TN_NOMFMA), each with its reduce kernel.
R="$GRAFT_REPO_ROOT"; cd /tmp && export TMPDIR=/tmp
for ex in "" "-DLPF_TN_NOLOAD" "-DLPF_TN_NOMFMA"; do
  touch $R/lpformer_amd/csrc/gemm_f32.hip; make -C $R/lpformer_amd/csrc EXTRA="$ex" > /dev/null 2>&1
  for w in 1; do
    rm -rf /tmp/tnp
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tnp -- python3 $R/tools/gemm_tn_bench.py > /dev/null 2>&1
    echo "[$ex]"; python3 $R/tools/kernel_median.py /tmp/tnp gemm_tn
  done
done
touch $R/lpformer_amd/csrc/gemm_f32.hip; make -C $R/lpformer_amd/csrc > /dev/null 2>&1
