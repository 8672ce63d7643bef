# round 3, batch 4e: adaptive tile order as the default (kernel-written tile costs): GPU suite + bench lines
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r4e; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu --tb=short 2>&1 | tail -12 > gpurun_out/r4e/pytest.log; tail -3 gpurun_out/r4e/pytest.log
for w in "north-star" "c2" "c5" "render" "render --tile-order static" "render --tile-order tail" "render --tile-order xcd" "north-star --quantiles 2"; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['value'], 'fwd', d['detail']['forward_ms'], 'bwd', d['detail']['backward_ms'])"
done
