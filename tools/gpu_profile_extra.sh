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
run_trace c4_rank python3 $REPO/bench.py --workload c4 --shards 8 --steps 3 --warmup 1 --no-cpu-baseline
run_trace c5_uniform python3 $REPO/bench.py --workload c5-uniform --steps 3 --warmup 1 --no-cpu-baseline
run_trace c5_block python3 $REPO/bench.py --workload c5-block --steps 5 --warmup 2 --no-cpu-baseline
run_trace c5_block_f64 python3 $REPO/bench.py --workload c5-block-f64 --steps 5 --warmup 2 --no-cpu-baseline
run_trace c1 python3 $REPO/bench.py --workload c1 --steps 20 --warmup 3 --no-cpu-baseline
run_trace poly python3 $REPO/tools/poly_timing.py
run_trace edd python3 $REPO/tools/edd_timing.py
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  echo "== pmc c5-uniform $C"
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_c5u_$N -o bench -- python3 $REPO/bench.py --workload c5-uniform --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_c5u_$N.json 2> $OUT/pmc_c5u_$N.err || { echo "pmc $C failed"; tail -3 $OUT/pmc_c5u_$N.err; }
done
find $OUT -name "*kernel_stats.csv"
