#!/bin/bash
# Bytes per phase call (run ON the GPU box from the repo root): tools/dev/phase_traffic.py under the FETCH_SIZE and WRITE_SIZE counters.
TAG=${1:-r3}
OUT=gpurun_out/phase_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 tools/dev/phase_traffic.py > $OUT/run_$c.log 2>&1 || exit 1
done
python3 tools/phase_traffic_summary.py $OUT
