#!/bin/bash
# Round-6 profile collection (GPU box): tools/r06_collect.sh <commit> [part ...]   parts: base traffic sq all
COMMIT=${1:-unknown}; shift
PARTS=${@:-base traffic sq all}
cd "$GRAFT_REPO_ROOT" || exit 1
for part in $PARTS; do
case $part in
base) bash tools/collect_profiles.sh r06 $COMMIT > gpurun_out/r06_collect.log 2>&1; tail -3 gpurun_out/r06_collect.log;;
traffic) bash tools/pmc_all_configs.sh r06 $COMMIT "ddi cora ppa citation2" > gpurun_out/r06_pmc_all.log 2>&1; tail -5 gpurun_out/r06_pmc_all.log;;
sq) bash tools/pmc_kernel.sh pair_rows_kernel collab > gpurun_out/r06_pmc_pair_rows.txt 2>&1
    bash tools/pmc_kernel.sh select4_kernel collab > gpurun_out/r06_pmc_select4.txt 2>&1
    bash tools/pmc_kernel.sh tail_chain_kernel collab > gpurun_out/r06_pmc_tail_chain.txt 2>&1
    python3 tools/pmc_sq_json.py gpurun_out/r06_pmc_sq_collab.json $COMMIT collab gpurun_out/pmc_pair_rows_kernel gpurun_out/pmc_select4_kernel gpurun_out/pmc_tail_chain_kernel;;
all) bash tools/all_configs.sh r06 > gpurun_out/r06_all_configs.log 2>&1; tail -30 gpurun_out/r06_all_configs.log;;
esac
done
