# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 r16): first pass at 4 waves per SIMD (RF_DELAUNAY_WAVES=4: 128 VGPRs) on the fast (bA) and the slow (base) build
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r
echo "-- r22 (kept: two-pass star_mark, hole list, sorted seeds, free_slot; the sweep compiled out; second pass with four blocks per trip)" >> gpurun_out/r/delaunay_stages.log
for v in base bA; do
  L=$R/radfoam_amd/libradfoam_hip_$v.so
  [ "$v" = "base" ] && L=$R/radfoam_amd/libradfoam_hip.so
  RADFOAM_HIP_LIB=$L timeout 600 python scripts/gpu_delaunay_stages.py 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r/delaunay_stages.log
done
tail -4 gpurun_out/r/delaunay_stages.log
