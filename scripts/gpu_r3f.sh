# Round 3, sixth GPU call: forward occupancy of the SH-3 instances on the flat batch (and on the 4K frame of config 5).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f
mkdir -p $O/ab
cd $R
VARIANTS="base d3w6 d3w7 d3w8" AB_STEPS=6 BENCH_EXTRA="--workload train-batch" bash scripts/gpu_ab.sh > $O/ab_train_batch.log 2>&1; cat $O/ab_train_batch.log
cp gpurun_out/ab/*.json $O/ab/
VARIANTS="base d3w6" AB_STEPS=5 BENCH_EXTRA="--workload c5" bash scripts/gpu_ab.sh > $O/ab_c5.log 2>&1; cat $O/ab_c5.log
