# round 3, batch x: evidence of the tree with the eager forward instances + staircase probe + shard simulation
R=$GRAFT_REPO_ROOT; cd $R
PMC_WORKLOADS="north-star train-batch" bash scripts/gpu_evidence.sh
mkdir -p $R/gpurun_out/r3x
timeout 300 python scripts/gpu_shard_probe.py > gpurun_out/r3x/shard_probe.jsonl 2> gpurun_out/r3x/probe.err
timeout 600 python scripts/gpu_shard_sim.py > gpurun_out/r3x/shard_simulation_one_gpu.json 2> gpurun_out/r3x/shard_sim.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3x/shard_simulation_one_gpu.json'))
for k,v in d['worlds'].items():
    print(k, 'max', v['max_rank_device_ms_without_collectives'], 'mean', v['mean_rank_device_ms_without_collectives'])
PY
