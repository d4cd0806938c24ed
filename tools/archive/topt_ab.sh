#!/bin/bash
# A/B of the candidate-list length of the pivoting sweeps; variants are built into /tmp and selected with
# SPR_HIP_LIBRARY (the shipped library is never overwritten)
set -e
cd "$(dirname "$0")/.."
for v in 8 16; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -DQR_TOPT_N=$v -c openmeasure_amd/csrc/qr_pivot.hip -o /tmp/qr_v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libspr_variant.so /tmp/qr_v.o $(ls build/csrc/*.o | grep -v qr_pivot) -Wl,-rpath,/opt/rocm/lib
  echo "== QR_TOPT=$v"
  SPR_HIP_LIBRARY=/tmp/libspr_variant.so python tools/placement_probe.py c3 2>&1 | grep -E "rep 2|class call [34]"
done
