# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 aj): two ranks on ONE GPU under gloo on the final sources: the two-rank GPU tests and the data-parallel loop as a functional run
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/aj
(timeout 900 python -m pytest tests/test_dist_training.py -m gpu -q -x 2>&1 | tail -5) > gpurun_out/aj/pytest_two_ranks.log; tail -2 gpurun_out/aj/pytest_two_ranks.log
(HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload train-loop --steps 60 --backend gloo --points 300000 2>gpurun_out/aj/loop2.err | tail -1) > gpurun_out/aj/train-loop_two_ranks_one_gpu_functional.json
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/aj/train-loop_two_ranks_one_gpu_functional.json")); det=d["detail"]
    print(d["value"], d["n_gpus"], det["world_size"], det["rays_per_rank"], det["last_exchange"], det["loss_first"], det["loss_last"], det["densification"], {k:v for k,v in det["ms_per_iteration"].items() if v>0.5})
except Exception as e:
    print("loop2 failed", e); print(open("gpurun_out/aj/loop2.err").read()[-2000:])
PY
