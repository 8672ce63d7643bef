# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 h): the rescans of contested cells as real functions (-DRF_OUTLINE_RESCANS=1) against inlined
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/h
(RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_outline.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5) > gpurun_out/h/pytest_outline.log; tail -3 gpurun_out/h/pytest_outline.log
for w in north-star c2 render c5 train-batch; do
  VARIANTS="base outline base outline" AB_STEPS=20 BENCH_EXTRA="--workload $w --no-repeated-frame" bash scripts/gpu_ab.sh 2>&1 | sed "s/^/$w /"
done > gpurun_out/h/outline_ab.log; cat gpurun_out/h/outline_ab.log
