#!/bin/bash
# Round-5 evidence beside collect_profiles.sh (GPU box, repo root): in-kernel stamps and ablation builds of select4 and of
# the pair-major attention behind it, the launch shapes of select4, the split-bf16 tail A/B.  -> gpurun_out/r05_*.txt
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
{
  echo "# select4_kernel<1024, 2>, collab-like: in-kernel stamps (tools/select4_stamps.sh), then ablation builds"
  bash tools/select4_stamps.sh "" collab 0
  for f in "-DS4_ABL_ONEPIECE" "-DS4_ABL_NOBUCKET"; do echo "== $f (timing only, wrong results)"; bash tools/select4_stamps.sh "$f" collab 0 | tail -11; done
  echo "== one block of 64 pairs per workgroup (threads = 1024)"; bash tools/select4_stamps.sh "" collab 1024 | tail -11
  echo "== ppa-like, default shape"; bash tools/select4_stamps.sh "" ppa 0 | tail -11
} > $O/r05_select4_stamps.txt 2>&1
{
  echo "# serial kernel times (HIP events, tools/fused_variants.py) by launch shape of select4: threads + 4096 * (blocks per workgroup - 1)"
  for t in 0 1024 512 4608 13312; do echo "threads=$t: $(LPF_SEL4_THREADS=$t timeout 300 python3 tools/fused_variants.py 2>&1 | tail -1 | cut -c1-300)"; done
  echo "# select3 (two launches, type-major) for comparison"
  echo "select_blocks=0: $(LPF_SELECT_BLOCKS=0 timeout 300 python3 tools/fused_variants.py 2>&1 | tail -1 | cut -c1-300)"
  echo "# dense tail: fp32 MFMAs / split-bf16 products"
  for sp in 0 1; do echo "tail_split=$sp: $(LPF_TAIL_SPLIT=$sp timeout 300 python3 tools/fused_variants.py 2>&1 | tail -1 | cut -c1-300)"; done
} > $O/r05_select4_shapes.txt 2>&1
{
  echo "# pair_rows_kernel (PT form behind select4), collab-like: per-wavefront stamps (tools/rows_stamps.py)"
  touch lpformer_amd/csrc/pair_rows.hip; make -C lpformer_amd/csrc EXTRA="-DPR_STAMPS" > /dev/null 2>&1
  LPF_STAMP_WG=1 python3 tools/rows_stamps.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|stddev" | head -30
  echo "== type-major form (select3) for comparison"
  LPF_SELECT_BLOCKS=0 python3 tools/rows_stamps.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|stddev" | head -9
  echo "== without detection / correction of flipped units (-DPR_ABL_NOFLIP, timing only)"
  touch lpformer_amd/csrc/pair_rows.hip; make -C lpformer_amd/csrc EXTRA="-DPR_STAMPS -DPR_ABL_NOFLIP" > /dev/null 2>&1
  python3 tools/rows_stamps.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|stddev" | head -9
  touch lpformer_amd/csrc/pair_rows.hip; make -C lpformer_amd/csrc > /dev/null 2>&1
} > $O/r05_rows_stamps.txt 2>&1
for t in split f32 split f32; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --tail $t --no-cpu-baseline --weights random --no-bf16 --no-kernel-timing 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; print('tail=$t', d['value'], d['ms_per_step'], d['ms_per_step_repeats'], c['launch_probe_ms_per_step'])"; done > $O/r05_tail_split_ab.txt 2>&1
tail -n 3 $O/r05_select4_stamps.txt $O/r05_select4_shapes.txt $O/r05_rows_stamps.txt $O/r05_tail_split_ab.txt
