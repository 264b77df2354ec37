#!/bin/bash
# diagnostic: times the dense kernel under the WAGG_DENSE_DBG knobs
for d in ${DBGS:-0 8 16}; do
  echo -n "DBG=$d "; WAGG_DENSE_DBG=$d python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print('kernel_ms', round(r['roofline']['kernel_ms_avg'],2), 'frac', round(r['roofline']['frac'],3))"
done
