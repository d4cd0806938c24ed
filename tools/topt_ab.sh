set -e
cd $GRAFT_REPO_ROOT
for v in 8 16; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -DQR_TOPT_N=$v -c openmeasure_amd/csrc/qr_pivot.hip -o /tmp/qr_v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o openmeasure_amd/libspr_hip.so /tmp/qr_v.o $(ls build/csrc/*.o | grep -v qr_pivot) -Wl,-rpath,/opt/rocm/lib
  echo "== QR_TOPT=$v"
  python tools/placement_probe.py c3 2>&1 | grep -E "rep 2|class call [34]"
done
