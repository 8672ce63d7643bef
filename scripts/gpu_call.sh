# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 t4): second pass with the waves of a block sharing queries (RF_DELAUNAY_COOP_WAVES x RF_DELAUNAY_COOP_GROUP), GPU tests of the triangulation
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/t
for cfg in "16 16" "16 8" "8 8" "4 4"; do
  set -- $cfg
  echo "== waves $1 group $2" >> gpurun_out/t/coop_groups.log
  RF_DELAUNAY_COOP_WAVES=$1 RF_DELAUNAY_COOP_GROUP=$2 RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_coopsec.so timeout 600 python scripts/gpu_delaunay_stages.py 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/t/coop_groups.log
done
cat gpurun_out/t/coop_groups.log
