# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 ah): the training loop over 3000 iterations on the final sources (round 4: 39.7 it/s on a scene at its optimum)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ah
(timeout 1500 python bench.py --workload train-loop --steps 3000 2>gpurun_out/ah/loop.err | tail -1) > gpurun_out/ah/train-loop_3000.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/ah/train-loop_3000.json")); det=d["detail"]
print(d["value"], d["unit"], {k:v for k,v in det["ms_per_iteration"].items() if v>0.3}, det.get("loss_first"), det.get("loss_last"), det.get("rebuilds"), det.get("densification"))
PY
