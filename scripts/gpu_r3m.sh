# Round 3: grouped flush of the image-path gradient cache (RF_FLUSH_GROUP row pairs per LDS round trip).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3m
mkdir -p $O/ab
cd $R
(RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_fl2.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "backward or autograd or full_frame" 2>&1 | tail -5) > $O/pytest_fl2.log; tail -1 $O/pytest_fl2.log
VARIANTS="base fl2 fl4 base fl2 fl4" AB_STEPS=12 bash scripts/gpu_ab.sh > $O/ab_north_star.log 2>&1; cat $O/ab_north_star.log
cp gpurun_out/ab/*.json $O/ab/
VARIANTS="base fl2 fl4" AB_STEPS=10 BENCH_EXTRA="--workload c2" bash scripts/gpu_ab.sh > $O/ab_c2.log 2>&1; cat $O/ab_c2.log
