# round 3, batch 4d: blocks take the tiles longest first / cheapest last (by the previous frame's hop counts)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r4d; cd $R
for w in north-star c2 c5; do
  for m in static xcd tail:1024 tail tail:3072 tail:4096; do
    timeout 400 python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline --tile-order $m 2>/dev/null | tail -1 > gpurun_out/r4d/${w}_$m.json
    python - $w $m <<'PY'
import json,sys
d=json.load(open('gpurun_out/r4d/%s_%s.json'%(sys.argv[1],sys.argv[2])))
print(sys.argv[1], sys.argv[2], 'Mrays/s', d['value'], 'fwd', d['detail']['forward_ms'], 'bwd', d['detail']['backward_ms'])
PY
  done
done | tee gpurun_out/r4d/summary.log
