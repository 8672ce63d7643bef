# Round 3, fifth GPU call: SH-row touch one hop ahead in the flat-batch backward; occupancy throttle of the flat-batch forward.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
(RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_pfrows.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "backward_parity or shuffled or autograd" 2>&1 | tail -5) > $O/pytest_pfrows.log; tail -1 $O/pytest_pfrows.log
VARIANTS="base pfrows" AB_STEPS=6 BENCH_EXTRA="--workload train-batch" bash scripts/gpu_ab.sh > $O/ab_train_batch.log 2>&1; cat $O/ab_train_batch.log
mkdir -p $O/ab_tb; cp gpurun_out/ab/*.json $O/ab_tb/
VARIANTS="base pfrows" AB_STEPS=6 BENCH_EXTRA="--workload train-batch --sh-degree 2" bash scripts/gpu_ab.sh > $O/ab_train_batch_sh2.log 2>&1; cat $O/ab_train_batch_sh2.log
for lds in 0 40000 53000 64000; do
  echo "forward throttle lds=$lds"
  RF_EXPERIMENT_FWD_LDS=$lds VARIANTS="envk" AB_STEPS=6 BENCH_EXTRA="--workload train-batch" bash scripts/gpu_ab.sh 2>&1 | tail -1
  cp gpurun_out/ab/envk.json $O/ab_tb/envk_lds$lds.json
done > $O/ab_fwd_throttle.log 2>&1; cat $O/ab_fwd_throttle.log
