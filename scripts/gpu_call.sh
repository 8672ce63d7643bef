# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 ab): stars dealt by previous neighbour count within chunks of 256 / 128 points
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/aa
for v in oc256 oc128; do
  echo "== $v" >> gpurun_out/aa/star_order_ab.log
  RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_$v.so timeout 300 python scripts/gpu_delaunay_stages.py 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/aa/star_order_ab.log
done
tail -4 gpurun_out/aa/star_order_ab.log
