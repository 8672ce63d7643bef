R=$GRAFT_REPO_ROOT; cd $R
for w in north-star c2; do echo "== $w"; VARIANTS="base fe4 fe5" BENCH_EXTRA="--workload $w" AB_STEPS=8 bash scripts/gpu_ab.sh 2>&1 | tail -3; done
