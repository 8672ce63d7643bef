# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 r6): star_mark in two passes (filter without stores, then flags), the hole as a list
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r
echo "-- r6" >> gpurun_out/r/delaunay_stages.log
for v in stage1 stage2 base nosweep; do
  L=$R/radfoam_amd/libradfoam_hip_$v.so
  [ "$v" = "base" ] && L=$R/radfoam_amd/libradfoam_hip.so
  RADFOAM_HIP_LIB=$L timeout 600 python scripts/gpu_delaunay_stages.py 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r/delaunay_stages.log
done
cat gpurun_out/r/delaunay_stages.log | tail -6
RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip.so timeout 600 python scripts/gpu_delaunay.py 500000 1 2000000 5 2>&1 | grep -v amdgpu.ids
