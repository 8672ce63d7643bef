# Round 3: cell records in front of the face blocks (RF_GEO_HEADERS) -- flat batch and image frame; head = the committed build.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3k
mkdir -p $O/ab
cd $R
for v in geohdr; do
(RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_binding.py -m gpu -q --tb=short 2>&1 | tail -5) > $O/pytest_$v.log; tail -1 $O/pytest_$v.log
done
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short 2>&1 | tail -5) > $O/pytest_base.log; tail -1 $O/pytest_base.log
VARIANTS="head base geohdr head base geohdr" AB_STEPS=12 bash scripts/gpu_ab.sh > $O/ab_north_star.log 2>&1; cat $O/ab_north_star.log
cp gpurun_out/ab/*.json $O/ab/
VARIANTS="head base geohdr" AB_STEPS=6 BENCH_EXTRA="--workload train-batch" bash scripts/gpu_ab.sh > $O/ab_tb.log 2>&1; cat $O/ab_tb.log
for v in head base geohdr; do cp gpurun_out/ab/$v.json $O/ab/tb_$v.json; done
VARIANTS="head base geohdr" AB_STEPS=6 BENCH_EXTRA="--workload c5" bash scripts/gpu_ab.sh > $O/ab_c5.log 2>&1; cat $O/ab_c5.log
VARIANTS="head base geohdr" AB_STEPS=10 BENCH_EXTRA="--workload render" bash scripts/gpu_ab.sh > $O/ab_render.log 2>&1; cat $O/ab_render.log
