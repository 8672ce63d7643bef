# round 3, batch 4a: trail replay with the prefetch one hop deeper (trail entry three hops ahead, cell record two)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r4a; cd $R
RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_deep.so timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "backward or short_and or autograd" 2>&1 | tail -2
for w in north-star c2 train-batch; do echo "== $w"; VARIANTS="base deep" BENCH_EXTRA="--workload $w" AB_STEPS=8 bash scripts/gpu_ab.sh 2>&1 | tail -2; done | tee gpurun_out/r4a/ab.log
for v in base deep; do
  L=$R/radfoam_amd/libradfoam_hip_$v.so; [ "$v" = "base" ] && L=$R/radfoam_amd/libradfoam_hip.so
  RADFOAM_HIP_LIB=$L timeout 300 python scripts/gpu_shard_probe.py --starts 0 464 --rows 16 128 > gpurun_out/r4a/probe_$v.jsonl 2>/dev/null
  echo $v; cut -c1-110 gpurun_out/r4a/probe_$v.jsonl
done
