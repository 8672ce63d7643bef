# Round 3: fp32 SH rows of odd pitch (d = 1, 3) read in place through unaligned 16-byte loads instead of repacked per step.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3n
mkdir -p $O/ab
cd $R
(RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_shinplace.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short 2>&1 | tail -5) > $O/pytest_shinplace.log; tail -1 $O/pytest_shinplace.log
VARIANTS="base shinplace base shinplace" AB_STEPS=6 BENCH_EXTRA="--workload train-batch" bash scripts/gpu_ab.sh > $O/ab_tb.log 2>&1; cat $O/ab_tb.log
VARIANTS="base shinplace base shinplace" AB_STEPS=6 BENCH_EXTRA="--workload c5" bash scripts/gpu_ab.sh > $O/ab_c5.log 2>&1; cat $O/ab_c5.log
