cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -x -q 2>&1 | tail -8
for mode in "" mask removed gpu; do
  for delta in 1 0; do
    [ -z "$mode" ] && [ "$delta" = 0 ] && continue
    [ "$mode" = removed ] && [ "$delta" = 0 ] && continue
    echo "train masked='$mode' delta=$delta: $(LPF_TRAIN_MASKED=$mode LPF_MASK_DELTA=$delta timeout 300 python3 tools/train_time.py 2>&1 | grep 'train step')"
  done
done 2>&1 | tee gpurun_out/r06_train_modes.txt
LPF_TRAIN_MASKED=mask LPF_TRAIN_PROFILE=1 timeout 600 python3 tools/train_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_train_profile_mask.txt
grep "train step" gpurun_out/r06_train_profile_mask.txt
VARIANTS=$'\n-DLPF_TC_GROUPS=2' SERIAL=1 bash tools/r06_ab.sh 2>&1 | tee gpurun_out/r06_ab2.txt
