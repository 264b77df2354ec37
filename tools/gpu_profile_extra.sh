#!/bin/bash
# rocprofv3 kernel-trace summaries of the widened rows: c5-block (tile-sparse dense form), fused
# tas_poly and Snyder degree days.  usage (GPU box): tools/gpu_profile_extra.sh <tag>
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_extra_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -o bench -- python3 $REPO/bench.py --workload c5-block --steps 5 --warmup 2 > $OUT/c5_bench.json 2> $OUT/c5.err || { echo c5 failed; tail -3 $OUT/c5.err; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/poly -o t -- python3 $REPO/tools/poly_timing.py > $OUT/poly.json 2> $OUT/poly.err || { echo poly failed; tail -3 $OUT/poly.err; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/edd -o t -- python3 $REPO/tools/edd_timing.py > $OUT/edd.json 2> $OUT/edd.err || { echo edd failed; tail -3 $OUT/edd.err; }
find $OUT -name "*kernel_stats.csv"
