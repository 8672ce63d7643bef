# round 3, batch r: forward with the first K face blocks of a cell requested at once (latency-bound small launches)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3r; cd $R
for v in base eager4 eager6 eager8; do
  L=$R/radfoam_amd/libradfoam_hip_$v.so; [ "$v" = "base" ] && L=$R/radfoam_amd/libradfoam_hip.so
  RADFOAM_HIP_LIB=$L timeout 300 python scripts/gpu_shard_probe.py --starts 0 464 --rows 16 64 128 136 > gpurun_out/r3r/probe_$v.jsonl 2> gpurun_out/r3r/probe_$v.err
  echo $v; python - $v <<'PY'
import json,sys
for l in open('gpurun_out/r3r/probe_%s.jsonl'%sys.argv[1]):
    d=json.loads(l); print('  ',d['first_row'],d['rows'],'fwd',d['forward_ms'],'bwd',d['backward_ms'])
PY
done
RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_eager6.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward_image or short_images or large_image" 2>&1 | tail -3
VARIANTS="base eager4 eager6 eager8" bash scripts/gpu_ab.sh 2>&1 | tail -5
