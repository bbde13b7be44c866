#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for st in 6 8 12; do for rows in on off; do
echo "streams $st rows $rows: $(timeout 600 python3 bench.py --streams $st --rows $rows --no-cpu-baseline --weights random --no-bf16 --no-kernel-timing --repeats 3 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['ms_per_step_repeats'], r['config']['launch'][:5])")"
done; done 2>&1 | tee gpurun_out/r04l_streams.txt
LPF_TRAIN_BS=8192 LPF_FUSED_ADAM=1 LPF_TRAIN_PROFILE=1 timeout 600 python3 tools/train_time.py > gpurun_out/r04l_train_profile.txt 2>&1
grep "train step" gpurun_out/r04l_train_profile.txt
