#!/bin/bash
# rocprofv3 kernel statistics of the training step (tools/train_time.py, collab-like, no override): per-kernel calls and
# average durations over the warm-up + 50 timed steps.
R="$GRAFT_REPO_ROOT"; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trk -- python3 $R/tools/train_time.py > /tmp/trk.log 2>&1
grep "^train step" /tmp/trk.log
f=$(find /tmp/trk -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out/trk
head -45 "$f" > $R/gpurun_out/trk/train_kernel_stats.csv
python3 $R/tools/kernel_median.py /tmp/trk _kernel | sort -k8 -n -r | head -40 > $R/gpurun_out/trk/train_kernel_medians.txt
