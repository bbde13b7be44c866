#!/bin/bash
# Tuning aid: launch shapes / gather depth of gcn_fused_kernel (threads per workgroup, workgroups per CU, neighbours per step)
cd "$GRAFT_REPO_ROOT" || exit 1
for ex in "" "-DLPF_GF_THREADS=512 -DLPF_GF_PER_CU=1 -DLPF_GF_NB=4" "-DLPF_GF_THREADS=512 -DLPF_GF_PER_CU=1 -DLPF_GF_NB=5" "-DLPF_GF_THREADS=256 -DLPF_GF_PER_CU=2 -DLPF_GF_NB=4" "-DLPF_GF_THREADS=768 -DLPF_GF_PER_CU=1 -DLPF_GF_NB=3" "-DLPF_GF_THREADS=1024 -DLPF_GF_PER_CU=1 -DLPF_GF_NB=2"; do
  touch lpformer_amd/csrc/gcn_fused.hip
  make -C lpformer_amd/csrc EXTRA="$ex" > /dev/null 2>&1 || echo "build failed"
  echo "[$ex] $(timeout 300 python3 tools/enc_time.py 2>&1 | tail -1)"
done
