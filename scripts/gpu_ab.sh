# A/B of kernel builds: RADFOAM_HIP_LIB variants, same bench
R=$GRAFT_REPO_ROOT
for v in $VARIANTS; do
  L=$R/radfoam_amd/libradfoam_hip_$v.so
  RADFOAM_HIP_LIB=$L timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline $BENCH_EXTRA 2>/dev/null | tail -1 > /tmp/o.json
  python - "$v" <<'PY'
import json,sys
d=json.load(open('/tmp/o.json')); w=d['detail']['walk']
print(sys.argv[1], 'Mrays/s', d['value'], 'fwd', d['detail']['forward_ms'], 'pack', d['detail']['foam_pack_ms'], 'bwd', d['detail']['backward_ms'], 'staged frac', round(w.get('lane_steps_staged_in_lds',0)/w['cells_scanned'],3), 'lane util', round(w['cells_scanned']/max(w.get('wave_steps',1),1)/64,3), 'wave_steps', w.get('wave_steps'))
PY
done
