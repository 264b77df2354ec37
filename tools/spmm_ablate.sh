#!/bin/bash
# Where does the entry-list kernel's time go?  Builds variants of the generated loop with parts left out
# (tools/gen_spmm_asm.py SPMM_ABL=...; results are WRONG, timing only) and times each on c5-uniform.
#   bash tools/spmm_ablate.sh build     here (hipcc cross-compiles): climate_toolbox_amd/lib/libwagg_abl_*.so
#   bash tools/spmm_ablate.sh run       on the GPU box
set -e
cd "$(dirname "$0")/.."
VARIANTS="${VARIANTS:-base nofma nolds noidx now nobfi nofma,nolds nolds,nobfi nofma,noidx}"
CS=climate_toolbox_amd/csrc
if [ "$1" = build ]; then
  make -C $CS >/dev/null
  for v in $VARIANTS; do
    tag=${v//,/_}
    if [ "$v" = base ]; then abl=""; else abl=$v; fi
    SPMM_ABL=$abl SPMM_OUT=$PWD/$CS/_obj/spmm_abl_$tag.inc python3 tools/gen_spmm_asm.py 2>/dev/null
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DWAGG_SPMM_ASM_INC="\"_obj/spmm_abl_$tag.inc\"" \
        -c $CS/wagg_spmm.hip -o $CS/_obj/spmm_abl_$tag.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o climate_toolbox_amd/lib/libwagg_abl_$tag.so \
        $CS/_obj/wagg_util.o $CS/_obj/wagg_host.o $CS/_obj/wagg_labels.o $CS/_obj/wagg_build.o $CS/_obj/wagg_sparse.o $CS/_obj/wagg_dense.o $CS/_obj/spmm_abl_$tag.o 2>/dev/null
    echo built $tag
  done
else
  for v in $VARIANTS; do
    tag=${v//,/_}
    echo -n "$tag "
    python3 bench.py --workload c5-uniform --diag-lib libwagg_abl_$tag.so --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | \
      python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print('kernel_ms', round(r['roofline']['kernel_ms_avg'],2))"
  done
fi
