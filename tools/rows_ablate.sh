#!/bin/bash
# pair_rows timing builds: -DPR_ABL_NOFLIP = neither detection nor correction of flipped hidden units (wrong results),
# -DPR_ABL_ALLDETECT = every entry looks at its units, as before the no-flip box (right results)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for ex in "" "-DPR_ABL_ALLDETECT" "-DPR_ABL_NOFLIP" "-DPR_ABL_NOZ" "-DPR_ABL_NOFLIP -DPR_ABL_NOZ"; do
  touch lpformer_amd/csrc/pair_rows.hip
  make -C lpformer_amd/csrc EXTRA="$ex" > /dev/null 2>&1 || { echo "[$ex] build failed"; continue; }
  python3 bench.py --gpus 1 --steps 40 --warmup 5 --rows on --launch plan --weights random --no-bf16 --no-cpu-baseline > gpurun_out/rowsabl_bench.log 2>&1
  echo "[$ex] $(tail -1 gpurun_out/rowsabl_bench.log | python3 tools/all_configs_fmt.py | head -2 | cut -c1-110 | tr '\n' ' ')"
done
touch lpformer_amd/csrc/pair_rows.hip; make -C lpformer_amd/csrc > /dev/null 2>&1
