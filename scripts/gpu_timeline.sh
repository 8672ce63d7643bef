mkdir -p gpurun_out
export RADFOAM_HIP_LIB=$GRAFT_REPO_ROOT/radfoam_amd/libradfoam_hip_timeline.so
timeout 300 python scripts/gpu_timeline.py 2>&1 | grep -v amdgpu.ids | tail -32
timeout 300 python scripts/gpu_timeline_bwd.py 2>&1 | grep -v amdgpu.ids | tail -42
