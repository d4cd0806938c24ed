#!/bin/bash
# builds diagnostic variants of libspr_hip.so with parts of a kernel removed and times them (outputs are wrong by
# construction; only the times matter).  The variants go to /tmp and are selected with SPR_HIP_LIBRARY: the shipped
# openmeasure_amd/libspr_hip.so is never touched.   usage: SRC=project VARIANTS="-DPROJ_ABLATE=0 ..." tools/ablate.sh "cells,F,m,r" ...
set -e
cd "$(dirname "$0")/.."
SRC=${SRC:-stats_gram}
for v in ${VARIANTS:-"-DGRAM_ABLATE=0" "-DGRAM_ABLATE=1" "-DGRAM_ABLATE=2"}; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude $v -c openmeasure_amd/csrc/$SRC.hip -o /tmp/sg_v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libspr_variant.so /tmp/sg_v.o $(ls build/csrc/*.o | grep -v "/$SRC.o") -Wl,-rpath,/opt/rocm/lib
  echo "== $v"
  SPR_HIP_LIBRARY=/tmp/libspr_variant.so python tools/kbench.py "$@" 2>&1 | grep -E "${SHOW:-cells|stats_gram|project}"
done
