# Round 3: grouped row emission of the flat-batch backward (RF_EMIT_ROWS = rows whose LDS reads are issued back to back).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3l
mkdir -p $O/ab
cd $R
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "backward_parity or shuffled or autograd" 2>&1 | tail -5) > $O/pytest_base.log; tail -1 $O/pytest_base.log
VARIANTS="emit1 emit2 base emit8 emit1 base" AB_STEPS=6 BENCH_EXTRA="--workload train-batch" bash scripts/gpu_ab.sh > $O/ab_train_batch.log 2>&1; cat $O/ab_train_batch.log
cp gpurun_out/ab/*.json $O/ab/
VARIANTS="emit1 base" AB_STEPS=6 BENCH_EXTRA="--workload train-batch --sh-degree 2" bash scripts/gpu_ab.sh > $O/ab_train_batch_sh2.log 2>&1; cat $O/ab_train_batch_sh2.log
