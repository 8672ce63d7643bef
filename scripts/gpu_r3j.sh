# Round 3: density-only segments of the image-path backward through a lock-free table instead of cache rows.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3j
mkdir -p $O/ab
cd $R
for v in dtab; do
(RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "backward or autograd or full_frame or geometry_only" 2>&1 | tail -5) > $O/pytest_$v.log; tail -1 $O/pytest_$v.log
done
VARIANTS="base dtab dtab1k base dtab" AB_STEPS=12 bash scripts/gpu_ab.sh > $O/ab_north_star.log 2>&1; cat $O/ab_north_star.log
cp gpurun_out/ab/*.json $O/ab/
VARIANTS="base dtab" AB_STEPS=10 BENCH_EXTRA="--workload c2" bash scripts/gpu_ab.sh > $O/ab_c2.log 2>&1; cat $O/ab_c2.log
VARIANTS="base dtab" AB_STEPS=10 BENCH_EXTRA="--quantiles 2" bash scripts/gpu_ab.sh > $O/ab_q2.log 2>&1; cat $O/ab_q2.log
