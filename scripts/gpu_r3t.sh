# round 3, batch t: eager face blocks on flat batches: K and waves per SIMD
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3t; cd $R
for w in "train-batch" "train-batch --sh-degree 2" "train-batch --quantiles 2"; do
  echo "== $w"; VARIANTS="base e6b e6c e8 e8d" BENCH_EXTRA="--workload $w" AB_STEPS=5 bash scripts/gpu_ab.sh 2>&1 | tail -5
done | tee gpurun_out/r3t/ab.log
