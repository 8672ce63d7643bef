cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > gpurun_out/ev/pytest_gpu.log; tail -2 gpurun_out/ev/pytest_gpu.log
VARIANTS="base" AB_STEPS=10 bash scripts/gpu_ab.sh
# forward-only A/B of the trail entry width (the 16-bit build's backward is meaningless: forward time only)
VARIANTS="base trail16" AB_STEPS=10 BENCH_EXTRA="--forward-only" bash scripts/gpu_ab.sh
mkdir -p gpurun_out/ab_fwdonly; cp gpurun_out/ab/base.json gpurun_out/ab_fwdonly/trail32.json; cp gpurun_out/ab/trail16.json gpurun_out/ab_fwdonly/trail16.json
bash scripts/gpu_ab_history.sh
