# full evidence run: GPU parity suite, smoke, bench (with cpu baseline), torchrun world-size-1 path,
# rocprof kernel stats, HBM traffic PMC passes
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5) > gpurun_out/pytest_gpu.log; tail -2 gpurun_out/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/smoke.log; tail -1 gpurun_out/smoke.log
(timeout 900 python bench.py 2>&1 | tail -1) > gpurun_out/bench.json
(timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_torchrun1.json
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench2m -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -2) > $R/gpurun_out/rocprof.log
cd $R
bash scripts/gpu_traffic.sh > gpurun_out/traffic.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print(d['value'],'Mrays/s; fwd',d['detail']['forward_ms'],'bwd',d['detail']['backward_ms'],'pack',d['detail']['foam_pack_ms'],'roofline',d['roofline'],'cpu',d.get('cpu_baseline'))
try:
    t=json.loads(open('gpurun_out/bench_torchrun1.json').read()); print('torchrun world=1:', t['value'], t['n_gpus'])
except Exception as e: print('torchrun path failed', e, open('gpurun_out/bench_torchrun1.json').read()[-500:])
PY
head -5 gpurun_out/prof/bench2m_kernel_stats.csv | cut -c1-150; tail -5 gpurun_out/traffic.log | cut -c1-250
