# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 al): second pass, subtrees dealt to a query's waves below the second level (base) against below the first (split0)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/al
(timeout 900 python -m pytest tests/test_delaunay.py -m gpu -q -x -k "hub or shell or configurations or equal_qhull" 2>&1 | tail -3) > gpurun_out/al/pytest.log; tail -2 gpurun_out/al/pytest.log
for v in base split0 base split0; do
  L=$R/radfoam_amd/libradfoam_hip_$v.so
  [ "$v" = "base" ] && L=$R/radfoam_amd/libradfoam_hip.so
  echo "== $v" >> gpurun_out/al/coop_split_ab.log
  RADFOAM_HIP_LIB=$L timeout 300 python scripts/gpu_delaunay_stages.py 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/al/coop_split_ab.log
done
cat gpurun_out/al/coop_split_ab.log
