# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 ai): the whole GPU suite and smoke() on the final tree (247 tests with the second-pass configurations)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ai
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/ai/pytest_gpu.log; tail -3 gpurun_out/ai/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/ai/smoke.log; tail -1 gpurun_out/ai/smoke.log
