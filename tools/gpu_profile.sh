#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (timed regions capped: a counter pass serialises every launch; the kernels are the same ones)
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --max-auto-steps 24 --warm-s 0.05 $*"
export WAGG_BENCH_TRACE=1
echo "== kernel trace"; 
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $REPO/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err || { echo trace failed; tail -5 $OUT/trace.err; exit 1; }
# (round 3: the default bench line carries every BASELINE config, so one set of passes covers every dominant kernel)
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  echo "== pmc $C"
  timeout -k 10 500 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o bench -- python3 $REPO/bench.py $ARGS > $OUT/pmc_$N.json 2> $OUT/pmc_$N.err || { echo "pmc $C failed"; tail -3 $OUT/pmc_$N.err; }
done
find $OUT -name "*.csv" | head -50
du -sh $OUT
