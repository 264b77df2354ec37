#!/bin/bash
# needs the diagnostic build: make -C climate_toolbox_amd/csrc diag  (the production library has no knobs)
# diagnostic: times the dense MFMA kernel under the WAGG_DENSE_DBG knobs (bit0 = no LDS-DMA in the
# k-loop, bit2 = no per-tile barrier); results are wrong with a knob set, only the time matters
for d in ${DBGS:-0 1 4 5}; do
  echo -n "DBG=$d "; WAGG_DENSE_DBG=$d python3 bench.py --diag-lib --steps 3 --warmup 1 --no-secondary --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print('kernel_ms', round(r['roofline']['kernel_ms_avg'],3), 'frac', round(r['roofline']['frac'],4), 'step_ms', round(r['ms_per_step'],3))"
done
