#!/bin/bash
# usage (GPU box): tools/pmc_kernel.sh <tag> <kernel-substring> "<counters>" [bench args...]
# one rocprofv3 --pmc pass of bench.py; prints the per-launch average of each counter for the
# kernels whose name contains the substring
set -o pipefail
TAG=$1; KSUB=$2; CNT=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc $CNT --output-format csv -d $OUT -o bench -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log || { echo failed; tail -5 $OUT/err.log; exit 1; }
python3 - "$(find $OUT -name 'bench_counter_collection.csv' | head -1)" "$KSUB" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r['Kernel_Name']: agg[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print("%-42s %-30s n=%d avg=%.6g" % (k[0],k[1],len(v),sum(v)/len(v)))
PY
