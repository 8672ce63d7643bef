# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 q): two ranks on one GPU under gloo (the data-parallel step with nothing replaced; the loop as a functional
# run), then the bench lines again so that they carry the roofline of the counters committed from the evidence run
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/q
(timeout 900 python -m pytest tests/test_dist_training.py -m gpu -q -x 2>&1 | tail -15) > gpurun_out/q/pytest_two_ranks.log; tail -5 gpurun_out/q/pytest_two_ranks.log
(HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload train-loop --steps 60 --backend gloo --points 300000 2>gpurun_out/q/loop2.err | tail -1) > gpurun_out/q/train-loop_two_ranks_one_gpu_functional.json
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/q/train-loop_two_ranks_one_gpu_functional.json")); det=d["detail"]
    print(d["value"], d["n_gpus"], d["data"][:60], det["world_size"], det["rays_per_rank"], det["last_exchange"], det["loss_first"], det["loss_last"], det["densification"], det["ms_per_iteration"])
except Exception as e:
    print("loop2 failed", e); print(open("gpurun_out/q/loop2.err").read()[-3000:])
PY
bash scripts/gpu_evidence.sh bench configs 2>&1 | grep -v "roofline\|loop:\|cpu:\|other:" | tail -25
