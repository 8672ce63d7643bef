# round 3, batch u: eager forward instances chosen by launch shape: tests, workloads, staircase, 8-way shard simulation
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3u; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu --tb=short 2>&1 | tail -15 > gpurun_out/r3u/pytest.log; tail -3 gpurun_out/r3u/pytest.log
for w in "north-star" "train-batch" "train-batch --sh-degree 2" "train-batch --quantiles 2" "c2"; do
  echo "== $w"; VARIANTS="base" BENCH_EXTRA="--workload $w" AB_STEPS=6 bash scripts/gpu_ab.sh 2>&1 | tail -1
done | tee gpurun_out/r3u/ab.log
timeout 300 python scripts/gpu_shard_probe.py --starts 0 464 --rows 16 64 128 135 > gpurun_out/r3u/probe.jsonl 2> gpurun_out/r3u/probe.err
cut -c1-110 gpurun_out/r3u/probe.jsonl
timeout 400 python scripts/gpu_shard_sim.py --worlds 4 8 --cuts balanced even > gpurun_out/r3u/shard_sim.json 2> gpurun_out/r3u/shard_sim.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3u/shard_sim.json'))
for k,v in d['worlds'].items():
    print(k, v['bounds'], 'max', v['max_rank_device_ms_without_collectives'], 'mean', v['mean_rank_device_ms_without_collectives'], 'scatter', v['scatter_all_ranks_ms'])
    for r in v['ranks']: print('   ', r['rows'], 'fwd', r['forward_ms'], 'bwd', r['backward_ms'], 'pack', r['pack_ms'], 'compact', r['compact_ms'])
PY
