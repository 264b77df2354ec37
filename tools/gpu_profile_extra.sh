#!/bin/bash
# rocprofv3 kernel-trace summaries of the workloads the default bench line does not cover: the c4 rank shard
# (T = 1,369 of 10,950 rows, dense form), c5 in both structures (entry lists / tile-sparse MFMA) and in fp64,
# c1, fused tas_poly and Snyder degree days; plus the HBM counters of the entry-list kernel.
# usage (GPU box): tools/gpu_profile_extra.sh <tag>
set -o pipefail
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_extra_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_trace() {   # name, then the program and its arguments
  local name=$1; shift
  echo "== trace $name"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o t -- "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -3 $OUT/$name.err; }
}
# (round 3: c1 / c4 / c5 are secondaries of the default bench line: tools/gpu_profile.sh covers them)
run_trace poly python3 $REPO/tools/poly_timing.py
run_trace edd python3 $REPO/tools/edd_timing.py
find $OUT -name "*kernel_stats.csv"
