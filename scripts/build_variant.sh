# usage: scripts/build_variant.sh NAME "-DFLAGS"  -> radfoam_amd/libradfoam_hip_NAME.so (all sources, same flags as build.py)
set -e
cd "$(dirname "$0")/.."
OUT=radfoam_amd/libradfoam_hip_$1.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-result $2 \
  -o $OUT radfoam_amd/csrc/rf_kernels.hip radfoam_amd/csrc/rf_scene_ops.hip radfoam_amd/csrc/rf_adjacency.hip radfoam_amd/csrc/rf_grad_exchange.hip radfoam_amd/csrc/rf_delaunay.hip radfoam_amd/csrc/rf_tile_prior.hip 2>&1 | grep -E "error" -A5 || true
echo built $OUT
