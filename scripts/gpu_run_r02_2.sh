cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > gpurun_out/ev/pytest_gpu.log; tail -2 gpurun_out/ev/pytest_gpu.log
VARIANTS="base" AB_STEPS=10 bash scripts/gpu_ab.sh
VARIANTS="base stage64 d3w4" BENCH_EXTRA="--workload train-batch" bash scripts/gpu_ab.sh
bash scripts/gpu_ab_history.sh
PMC_WORKLOADS="train-batch" bash scripts/gpu_evidence.sh prof pmc | grep "train-batch\]"
