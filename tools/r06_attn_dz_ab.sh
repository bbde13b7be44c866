#!/bin/bash
# Kernel medians of the attention stage's training kernels under rocprofv3 (tools/train_time.py, collab-like):
# pair_attn_train_fwd / _bwd, the by-node sum of dK (segment_sum_kernel), the endpoint scatter.
R="$GRAFT_REPO_ROOT"; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ab; rocprofv3 --kernel-trace --output-format csv -d /tmp/ab -- python3 $R/tools/train_time.py > /tmp/ab.log 2>&1
grep "^train step" /tmp/ab.log | cut -c60-200
for k in pair_attn_train segment_sum pair_scatter radix onesweep; do python3 $R/tools/kernel_median.py /tmp/ab $k; done
