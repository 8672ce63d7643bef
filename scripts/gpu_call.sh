# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 a): the GPU suite on the data-parallel tree, the default line without extras, the loop on one GPU and
# through the launcher path (--gpus 1 under torch.distributed.run is world 1: no exchange)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/a
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > gpurun_out/a/pytest_gpu.log; tail -3 gpurun_out/a/pytest_gpu.log
timeout 400 python bench.py --no-cpu-baseline --no-other-workloads 2>gpurun_out/a/bench.err | tail -1 > gpurun_out/a/bench_default.json
timeout 500 python bench.py --workload train-loop --steps 200 2>gpurun_out/a/loop.err | tail -1 > gpurun_out/a/loop200.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/a/bench_default.json")); print(d["value"], d["detail"]["forward_ms"], d["detail"]["backward_ms"], d["detail"].get("value_repeated_frame"))
d = json.load(open("gpurun_out/a/loop200.json")); print(d["value"], "it/s", d["detail"]["ms_per_iteration"])
PY
