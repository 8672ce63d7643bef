# round 3, batch q: 32x8 blocks for small image launches: parity tests, staircase probe, 8-way shard simulation
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3q; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_blocks or backward_parity or forward_image or large_image" --tb=short 2>&1 | tail -30 > gpurun_out/r3q/pytest.log
cat gpurun_out/r3q/pytest.log | tail -5
timeout 300 python scripts/gpu_shard_probe.py --starts 0 464 --rows 128 135 136 144 > gpurun_out/r3q/probe.jsonl 2> gpurun_out/r3q/probe.err
cut -c1-120 gpurun_out/r3q/probe.jsonl
timeout 400 python scripts/gpu_shard_sim.py --worlds 8 --cuts capped even > gpurun_out/r3q/shard_sim_8.json 2> gpurun_out/r3q/shard_sim.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3q/shard_sim_8.json'))
for k,v in d['worlds'].items():
    print(k, v['bounds'], 'max', v['max_rank_device_ms_without_collectives'], 'mean', v['mean_rank_device_ms_without_collectives'], 'scatter', v['scatter_all_ranks_ms'])
    for r in v['ranks']: print('   ', r)
PY
