# full evidence run: GPU parity suite, smoke, bench (with cpu baseline), rocprof kernel stats
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5) > gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/smoke.log; tail -1 gpurun_out/smoke.log
(timeout 900 python bench.py 2>&1 | tail -1) > gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench2m -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -2) > $R/gpurun_out/rocprof.log
cd $R
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print(d['value'],'Mrays/s; fwd',d['detail']['forward_ms'],'bwd',d['detail']['backward_ms'],'roofline',d['roofline'],'cpu',d.get('cpu_baseline'))
PY
head -6 gpurun_out/prof/bench2m_kernel_stats.csv | cut -c1-160
