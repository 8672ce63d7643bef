# round 3, batch y: 1/2/4/8-way row-sharded forward of BASELINE config 5 (4 M points, 4K, SH 3), one rank after the other on one GPU
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3y; cd $R
timeout 900 python scripts/gpu_shard_sim.py --points 4000000 --seed 4 --sh-degree 3 --width 3840 --height 2160 --forward-only > gpurun_out/r3y/shard_simulation_c5_forward_only.json 2> gpurun_out/r3y/err.log
tail -3 gpurun_out/r3y/err.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3y/shard_simulation_c5_forward_only.json'))
for k,v in d['worlds'].items():
    print(k, v['bounds'], 'max', v['max_rank_device_ms_without_collectives'], 'mean', v['mean_rank_device_ms_without_collectives'], [r['forward_ms'] for r in v['ranks']])
PY
