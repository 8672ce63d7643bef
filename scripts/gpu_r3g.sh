# Round 3, seventh GPU call: mode 4 at 4 waves/SIMD (quarter-wave staging so that four blocks fit a CU's LDS).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
mkdir -p $O/ab
cd $R
(RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_st16w4.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "backward_parity or shuffled or autograd" 2>&1 | tail -5) > $O/pytest_st16w4.log; tail -1 $O/pytest_st16w4.log
VARIANTS="base st16 st16w4" AB_STEPS=6 BENCH_EXTRA="--workload train-batch" bash scripts/gpu_ab.sh > $O/ab_train_batch.log 2>&1; cat $O/ab_train_batch.log
cp gpurun_out/ab/*.json $O/ab/
