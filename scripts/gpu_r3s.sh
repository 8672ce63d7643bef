# round 3, batch s: eager face blocks across the workloads
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3s; cd $R
for w in train-batch c5 render c2 north-star; do
  echo "== $w"; VARIANTS="base e6a e6b e7" BENCH_EXTRA="--workload $w" AB_STEPS=5 bash scripts/gpu_ab.sh 2>&1 | tail -4
done | tee gpurun_out/r3s/ab.log
