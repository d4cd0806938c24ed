#!/bin/bash
# builds diagnostic variants of libspr_hip.so with parts of the Gram kernel removed and times them
# (outputs are wrong by construction; only the times matter).  usage: tools/ablate.sh "cells,F,m,r" ...
set -e
cd "$(dirname "$0")/.."
cp openmeasure_amd/libspr_hip.so /tmp/libspr_full.so
for v in ${VARIANTS:-"-DGRAM_ABLATE=0" "-DGRAM_ABLATE=1" "-DGRAM_ABLATE=2"}; do
  SRC=${SRC:-stats_gram}
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude $v -c openmeasure_amd/csrc/$SRC.hip -o /tmp/sg_v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o openmeasure_amd/libspr_hip.so /tmp/sg_v.o $(ls build/csrc/*.o | grep -v $SRC) -Wl,-rpath,/opt/rocm/lib
  echo "== $v"
  python tools/kbench.py "$@" 2>&1 | grep -E "cells|stats_gram|project"
done
cp /tmp/libspr_full.so openmeasure_amd/libspr_hip.so
