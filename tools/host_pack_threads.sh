set -e
D=$(mktemp -d)
python3 tools/host_gather_probe.py $D
/opt/rocm/bin/hipcc -O2 -mavx2 -pthread --offload-arch=gfx950 -o $D/host_gather tools/micro/host_gather.cpp
for th in 10 12 14 15 16; do echo "== fp32 quads, $th threads"; $D/host_gather $D/runs_4_g16.bin 365 1036800 4 $th 1 4; done
for th in 12 14 16; do echo "== fp64 quads, $th threads"; $D/host_gather $D/runs_8_g32.bin 365 1036800 8 $th 1 4; done
