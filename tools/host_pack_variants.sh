#!/bin/bash
# Which copy loop for the packing threads of the lines-only / quads-only host path?  Packing + copy pipelined (ring of four,
# 12 threads) with memcpy (nt 0), 16-byte loads + streaming stores (nt 1: what the library did through round 6's first half)
# and 32-byte AVX2 loads + streaming stores (nt 2), for the quads (16 B) and the whole lines (128 B) of c2-real, fp32 and fp64.
set -e
D=$(mktemp -d)
python3 tools/host_gather_probe.py $D
/opt/rocm/bin/hipcc -O2 -mavx2 -pthread --offload-arch=gfx950 -o $D/host_gather tools/micro/host_gather.cpp
for g in 16 128; do for nt in 1 0 2 3 4; do
  echo "== fp32 granule $g B, 12 threads, nt $nt"; $D/host_gather $D/runs_4_g$g.bin 365 1036800 4 12 $nt 4
done; done
for nt in 1 3 4; do echo "== fp64 granule 32 B (quads), 12 threads, nt $nt"; $D/host_gather $D/runs_8_g32.bin 365 1036800 8 12 $nt 4; done
rm -rf $D
