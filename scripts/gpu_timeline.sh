# per-block timelines (RF_EXPERIMENT_TIMELINE build) and cache counters (RF_EXPERIMENT_COUNTERS build):
#   scripts/build_variant.sh timeline -DRF_EXPERIMENT_TIMELINE; scripts/build_variant.sh counters -DRF_EXPERIMENT_COUNTERS
mkdir -p gpurun_out
export RADFOAM_HIP_LIB=$GRAFT_REPO_ROOT/radfoam_amd/libradfoam_hip_timeline.so
timeout 300 python scripts/gpu_timeline.py 2>&1 | grep -v amdgpu.ids | tail -32
timeout 300 python scripts/gpu_timeline_bwd.py 2>&1 | grep -v amdgpu.ids | tail -42
export RADFOAM_HIP_LIB=$GRAFT_REPO_ROOT/radfoam_amd/libradfoam_hip_counters.so
if [ -f $RADFOAM_HIP_LIB ]; then timeout 300 python scripts/gpu_timeline_bwd.py 2>&1 | grep "lit wave\|flush:"; fi
