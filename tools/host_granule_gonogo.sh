#!/bin/bash
# VERDICT r5 Next #8: go / no-go for finer PCIe granules of the lines-only host path.  Packers + copy, pipelined through a ring of
# four pieces (what wagg_host.hip does), for 16 / 32 / 64 / 128-byte granules of the c2-real table; fp32 (T = 365) and fp64.
# usage (GPU box): bash tools/host_granule_gonogo.sh > gpurun_out/host_granule.txt
set -e
D=$(mktemp -d)
python3 tools/host_gather_probe.py $D
/opt/rocm/bin/hipcc -O2 -mavx2 -pthread --offload-arch=gfx950 -o $D/host_gather tools/micro/host_gather.cpp
for th in 12 8; do
  for g in 128 64 32 16; do
    echo "== fp32 granule $g B, threads $th, streaming stores, ring of 4"
    $D/host_gather $D/runs_4_g$g.bin 365 1036800 4 $th 1 4
  done
done
for g in 128 64 32; do
  echo "== fp64 granule $g B, threads 12, streaming stores, ring of 4"
  $D/host_gather $D/runs_8_g$g.bin 365 1036800 8 12 1 4
done
rm -rf $D
