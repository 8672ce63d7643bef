# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 ae): the walk's child order from the parent's box (no load of the left child's bound before descending) against the committed walk (head)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/ae
for v in base head base head; do
  L=$R/radfoam_amd/libradfoam_hip_$v.so
  [ "$v" = "base" ] && L=$R/radfoam_amd/libradfoam_hip.so
  RADFOAM_HIP_LIB=$L timeout 300 python scripts/gpu_delaunay_stages.py 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/ae/child_order_ab.log
done
cat gpurun_out/ae/child_order_ab.log
