# Round 3, fourth GPU call: barrier-free mode 4 (lock-free write-back tables), one-instruction row emission.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -80) > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
VARIANTS="base noatomics" AB_STEPS=6 BENCH_EXTRA="--workload train-batch" bash scripts/gpu_ab.sh > $O/ab_train_batch.log 2>&1; cat $O/ab_train_batch.log
mkdir -p $O/ab_tb; cp gpurun_out/ab/*.json $O/ab_tb/
VARIANTS="base" AB_STEPS=6 BENCH_EXTRA="--workload train-batch --sh-degree 2" bash scripts/gpu_ab.sh > $O/ab_train_batch_sh2.log 2>&1; cat $O/ab_train_batch_sh2.log
VARIANTS="base" AB_STEPS=6 BENCH_EXTRA="--workload train-batch --quantiles 2" bash scripts/gpu_ab.sh > $O/ab_train_batch_q2.log 2>&1; cat $O/ab_train_batch_q2.log
