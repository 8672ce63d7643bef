# round 3, batch w: flat-batch replay: colour row requested a hop ahead, four table updates batched
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3w; cd $R
for w in "train-batch" "train-batch --sh-degree 2"; do
  echo "== $w"; VARIANTS="base pre0 tb0" BENCH_EXTRA="--workload $w" AB_STEPS=6 bash scripts/gpu_ab.sh 2>&1 | tail -3
done | tee gpurun_out/r3w/ab.log
RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_sections.so timeout 300 python scripts/gpu_sections.py 2> gpurun_out/r3w/sec3.err | tee gpurun_out/r3w/sections_sh3.json
RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_pre0.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "backward_parity and 4" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "backward_parity" 2>&1 | tail -2
