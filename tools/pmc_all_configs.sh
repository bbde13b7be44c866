#!/bin/bash
# HBM traffic counters per config (GPU box, repo root): two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only)
# of the serial bench per config -> gpurun_out/<tag>_pmc_traffic_<config>.json  (copy into profiles/)
TAG=${1:-r05}
COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
mkdir -p $OUT
for c in ${3:-collab ddi cora ppa citation2}; do
  for P in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$c/$P -- python3 bench.py --config $c --launch eager --streams 1 --weights random --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline --no-kernel-timing --no-bf16 > /dev/null 2>&1
  done
  python3 tools/pmc_traffic.py $OUT/${TAG}_pmc_$c/FETCH_SIZE $OUT/${TAG}_pmc_$c/WRITE_SIZE $OUT/${TAG}_pmc_traffic_$c.json $COMMIT $c > $OUT/${TAG}_pmc_traffic_$c.txt 2>&1
  rm -rf $OUT/${TAG}_pmc_$c
  echo "== $c"; cat $OUT/${TAG}_pmc_traffic_$c.txt
done
