# quick check on the GPU box: parity suite (stop at first failure) + the default bench line without the CPU baseline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > gpurun_out/ev/pytest_gpu.log; tail -3 gpurun_out/ev/pytest_gpu.log
VARIANTS="base $VARIANTS" AB_STEPS=${AB_STEPS:-10} bash scripts/gpu_ab.sh
