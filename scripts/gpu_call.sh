# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
(timeout 1500 python bench.py --workload train-loop --steps 3000 2>$O/loop.err | tail -1) > $O/bench_train-loop_3000.json; tail -c 300 $O/loop.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4n/bench_train-loop_3000.json")); print(d["value"], d["unit"], d["ms_per_step"]); print(json.dumps(d["detail"]["ms_per_iteration"])); print(d["detail"]["rebuilds"], d["detail"]["densification"], d["detail"]["wall_seconds"]); print(json.dumps(d["detail"]["ms_per_call"]))
PY
