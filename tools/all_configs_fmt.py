"""Formats one bench.py JSON line (stdin) for tools/all_configs.sh."""
import json
import sys

d = json.loads(sys.stdin.read())
print(d["value"], d["ms_per_step"], d.get("ms_per_step_repeats"), d["config"]["workload"])
print({k: v["ms_per_step"] for k, v in list(d.get("kernels", {}).items())[:6]})
bm = d.get("bf16_mode") or {}
print("bf16:", bm.get("value"), bm.get("max_abs_logit_diff_vs_f32"), "encoder ms f32 / bf16:", d.get("encoder_ms"),
      bm.get("encoder_ms"))
