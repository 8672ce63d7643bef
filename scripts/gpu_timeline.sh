mkdir -p gpurun_out
RADFOAM_HIP_LIB=$GRAFT_REPO_ROOT/radfoam_amd/libradfoam_hip_timeline.so timeout 300 python scripts/gpu_timeline.py 2>&1 | tail -40
