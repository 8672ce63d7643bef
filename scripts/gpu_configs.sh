# other BASELINE configs that fit one GPU and a cached foam: C2 (500k points, 1080p, SH 2, fwd+bwd)
# and the north-star foam forward-only
mkdir -p gpurun_out
(timeout 600 python bench.py --points 500000 --seed 1 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_c2.json
(timeout 600 python bench.py --forward-only --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_fwd.json
python - <<'PY'
import json
for f in ['gpurun_out/bench_c2.json','gpurun_out/bench_fwd.json']:
    try:
        d=json.loads(open(f).read()); print(f, d['value'], d['unit'], 'fwd', d['detail']['forward_ms'], 'bwd', d['detail']['backward_ms'], 'pack', d['detail']['foam_pack_ms'], 'cells/ray', d['detail']['mean_cells_per_ray'])
    except Exception as e: print(f, 'failed', e, open(f).read()[-300:])
PY
