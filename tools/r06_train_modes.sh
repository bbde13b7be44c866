#!/bin/bash
# Training step (tools/train_time.py) with the per-batch overrides of src/train/train_model.py:38-56 in every form:
#   ""       no override            mask     adj_mask only (the OGB scripts)      removed   lpformer_amd.RemovedEdges(edges)
#   gpu      adj_prop + adj_mask (--mask-input)        delta = 0: every override a graph of its own (rounds 2-5)
#   removed_both   RemovedEdges(edges) as adj_mask AND adj_prop
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for cfg in ${CFGS:-collab}; do
for mode in "" mask removed gpu removed_both; do
  for delta in 1 0; do
    [ -z "$mode" ] && [ "$delta" = 0 ] && continue
    [ "$mode" = removed ] && [ "$delta" = 0 ] && continue
    [ "$mode" = removed_both ] && [ "$delta" = 0 ] && continue
    echo "$cfg masked='$mode' delta=$delta: $(LPF_CFG=$cfg LPF_TRAIN_BS=${TRAIN_BS:-8192} LPF_TRAIN_MASKED=$mode LPF_MASK_DELTA=$delta timeout 600 python3 tools/train_time.py 2>&1 | grep '^train step')"
  done
done
done
