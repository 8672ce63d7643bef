mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
(timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $BENCH_EXTRA 2>&1 | tail -1) > gpurun_out/bench.log
python - <<'PY'
import json
for l in open('gpurun_out/bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'],'Mrays/s fwd', d['detail']['forward_ms'], 'bwd', d['detail']['backward_ms'], 'pack', d['detail']['foam_pack_ms'], 'frac', d['roofline']['frac'])
PY
