#!/bin/bash
# launch path chosen by the bench's probe, per config
cd "$GRAFT_REPO_ROOT" || exit 1
for c in ${@:-collab ppa citation2}; do
echo "[$c] $(timeout 900 python3 bench.py --config $c --no-cpu-baseline --repeats 3 2>&1 | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["launch"][:12], d["config"]["launch_probe_ms_per_step"], "bf16", d["bf16_mode"]["value"] if d.get("bf16_mode") and "value" in d["bf16_mode"] else None)')"
done
