# round 3, batch z: wave-clock shares of the image path's replay (backward mode 3) on the north-star frame
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3z; cd $R
WORKLOAD=north-star RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_sections.so timeout 300 python scripts/gpu_sections.py 2> gpurun_out/r3z/err.log | tee gpurun_out/r3z/sections_mode3_north-star.json
tail -2 gpurun_out/r3z/err.log
