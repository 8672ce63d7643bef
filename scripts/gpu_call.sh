# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 ac): the training loop over 1000 iterations on the final sources (round 5: 33.2 it/s)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ac
(timeout 1500 python bench.py --workload train-loop --steps 1000 2>gpurun_out/ac/loop.err | tail -1) > gpurun_out/ac/train-loop_1000.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/ac/train-loop_1000.json")); det=d["detail"]
print(d["value"], d["unit"], {k:v for k,v in det["ms_per_iteration"].items() if v>0.3}, det.get("loss_first"), det.get("loss_last"), det.get("rebuilds"))
PY
