#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -C lpformer_amd/csrc > /dev/null 2>&1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04j_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04j_tests.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/r04j_tests.log | tail -12
timeout 900 python3 bench.py > gpurun_out/r04j_bench.log 2>&1
echo "bench rc=$?"
tail -1 gpurun_out/r04j_bench.log | python3 tools/all_configs_fmt.py
tail -1 gpurun_out/r04j_bench.log | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('config', {k: r['config'][k] for k in ('attention_form','attention_form_probe_ms_per_step','launch','launch_probe_ms_per_step','flips_per_entry')})
print('trained', r['trained_weights'])
print('pair_stage', r['pair_stage'], 'encoder', r['encoder_ms'], r['node_keys_ms'], r['value_incl_encoder'])
"
timeout 900 python3 bench.py --steps 20 --no-cpu-baseline --weights random --no-bf16 2>/dev/null | tail -1 | python3 tools/all_configs_fmt.py
