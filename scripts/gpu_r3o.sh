# Round 3: occupancy target of the quantile instances of the forward (the training loop's call pattern).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3o
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -6) > $O/pytest_gpu.log; tail -1 $O/pytest_gpu.log
VARIANTS="base oth5 oth6 base oth5 oth6" AB_STEPS=10 BENCH_EXTRA="--quantiles 2" bash scripts/gpu_ab.sh > $O/ab_ns_q2.log 2>&1; cat $O/ab_ns_q2.log
VARIANTS="base oth5 oth6" AB_STEPS=6 BENCH_EXTRA="--workload train-batch --quantiles 2" bash scripts/gpu_ab.sh > $O/ab_tb_q2.log 2>&1; cat $O/ab_tb_q2.log
