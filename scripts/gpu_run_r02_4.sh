cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/small
# host overhead check: one eighth of the frame per step (what a rank of an 8-GPU strong-scaling run traces)
for h in 136 272; do
  timeout 300 python bench.py --height $h --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/small/h$h.json
  python - gpurun_out/small/h$h.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); t=d['detail']
print(sys.argv[1], 'ms/step', d['ms_per_step'], 'kernels fwd+bwd+pack', round(t['forward_ms']+t['backward_ms']+t['foam_pack_ms'],3), t['forward_ms'], t['backward_ms'], t['foam_pack_ms'])
PY
done
# world-size-1 exercise of the sharded path under the launcher (exchange code is a no-op at world 1)
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
