# round 3, batch v: wave-clock shares of the flat-batch replay's sections
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3v; cd $R
RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_sections.so timeout 300 python scripts/gpu_sections.py 2> gpurun_out/r3v/sec3.err | tee gpurun_out/r3v/sections_sh3.json
SH=2 RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_sections.so timeout 300 python scripts/gpu_sections.py 2> gpurun_out/r3v/sec2.err | tee gpurun_out/r3v/sections_sh2.json
tail -3 gpurun_out/r3v/sec3.err
timeout 400 python scripts/gpu_shard_sim.py --worlds 8 --cuts balanced even > gpurun_out/r3v/shard_sim.json 2> gpurun_out/r3v/shard_sim.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3v/shard_sim.json'))
for k,v in d['worlds'].items():
    print(k, v['bounds'], 'max', v['max_rank_device_ms_without_collectives'], 'mean', v['mean_rank_device_ms_without_collectives'], 'scatter', v['scatter_all_ranks_ms'])
    for r in v['ranks']: print('   ', r['rows'], 'fwd', r['forward_ms'], 'bwd', r['backward_ms'], 'pack', r['pack_ms'], 'compact', r['compact_ms'])
PY
