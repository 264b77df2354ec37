#!/bin/bash
# needs the diagnostic build: make -C climate_toolbox_amd/csrc diag  (the production library has no knobs)
# diagnostic: times the sparse gather kernel under the WAGG_SPARSE_DBG knobs
for d in ${DBGS:-0 1 2 3 4 6 7}; do
  echo -n "DBG=$d "; WAGG_SPARSE_DBG=$d python3 bench.py --diag-lib --workload ${WL:-c2-real} --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print('kernel_ms', round(r['roofline']['kernel_ms_avg'],4), 'step_ms', round(r['ms_per_step'],4))"
done
