mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
echo "=== bench 2M mode2" > gpurun_out/bench.log
(timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1) >> gpurun_out/bench.log
echo "=== bench 2M mode1" >> gpurun_out/bench.log
(timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --backward-mode 1 2>&1 | tail -1) >> gpurun_out/bench.log
python - <<'PY'
import json
for l in open('gpurun_out/bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'],'Mrays/s', d['detail']['forward_ms'], d['detail']['backward_ms'], d['detail']['foam_pack_ms'], d['roofline'])
    else: print(l.strip())
PY
