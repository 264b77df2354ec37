#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> "<counters>" [bench args...]   (one PMC pass, summary printed)
set -o pipefail
TAG=$1; CNT=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc $CNT --output-format csv -d $OUT -o bench -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log || { echo failed; tail -5 $OUT/err.log; exit 1; }
python3 - $OUT/bench_counter_collection.csv <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows: agg[(r['Kernel_Name'][:48], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    if any(s in k[0] for s in ('mfma','sparse_gather','reduce')): print("%-50s %-32s n=%d avg=%.6g" % (k[0],k[1],len(v),sum(v)/len(v)))
PY
