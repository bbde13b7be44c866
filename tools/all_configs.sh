#!/bin/bash
# Step time, kernel shares and rooflines of every synthetic config (GPU box, repo root)
#   -> gpurun_out/<tag>_all_configs.txt (readable) and gpurun_out/<tag>_all_configs.json (per config: value, ms_per_step,
#      rooflines, pair_stage_hbm_frac, index_bytes, setup_s, kernels)
TAG=${1:-r05}
OUT=gpurun_out/${TAG}_all_configs.txt
mkdir -p gpurun_out
: > $OUT
for c in ${2:-collab ddi cora ppa citation2}; do
  echo "== $c" >> $OUT
  timeout 900 python3 bench.py --config $c --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 > gpurun_out/${TAG}_line_$c.json
  python3 tools/all_configs_fmt.py < gpurun_out/${TAG}_line_$c.json >> $OUT 2>&1
done
python3 - <<PY
import json
out = {}
for c in "${2:-collab ddi cora ppa citation2}".split():
    try:
        d = json.load(open("gpurun_out/${TAG}_line_%s.json" % c))
    except Exception as exc:
        out[c] = {"error": str(exc)}
        continue
    out[c] = {k: d.get(k) for k in ("value", "unit", "ms_per_step", "ms_per_step_repeats", "roofline", "rooflines",
                                    "pair_stage_hbm_frac", "pair_stage", "kernels", "encoder_ms", "node_keys_ms",
                                    "value_incl_encoder", "trained_weights")}
    out[c]["workload"] = d["config"]["workload"]
    out[c]["launch"] = d["config"]["launch"]
    out[c]["launch_probe_ms_per_step"] = d["config"].get("launch_probe_ms_per_step")
    out[c]["attention_form"] = d["config"].get("attention_form")
    out[c]["selection_form"] = d["config"].get("selection_form")
    out[c]["setup_s"] = d.get("setup_s")
    out[c]["index_bytes"] = (d.get("setup_s") or {}).get("index_bytes")
json.dump(out, open("gpurun_out/${TAG}_all_configs.json", "w"), indent=1)
PY
cat $OUT
