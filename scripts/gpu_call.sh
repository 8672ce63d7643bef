# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 d): device-built tile orders + the coherence gate: GPU suite, the A/B script on both scenes, the default line
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/d
(timeout 900 python -m pytest tests -m gpu -q -x -rs 2>&1 | tail -12) > gpurun_out/d/pytest_gpu.log; tail -6 gpurun_out/d/pytest_gpu.log
timeout 600 python scripts/gpu_tile_prior.py --views 8 --res 32 --rules xcd > gpurun_out/d/tile_orders_north_star.json 2>gpurun_out/d/e1.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/d/tile_orders_north_star.json"))
for k,v in d["modes"].items(): print(k, {m: x["mean_ms"] for m,x in v.items()})
PY
tail -3 gpurun_out/d/e1.err
timeout 600 python scripts/gpu_tile_prior.py --asymmetric --views 8 --res 32 --rules xcd > gpurun_out/d/tile_orders_asymmetric.json 2>gpurun_out/d/e2.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/d/tile_orders_asymmetric.json"))
for k,v in d["modes"].items(): print(k, {m: x["mean_ms"] for m,x in v.items()})
PY
tail -3 gpurun_out/d/e2.err
timeout 600 python bench.py --no-cpu-baseline --no-other-workloads 2>gpurun_out/d/bench.err | tail -1 > gpurun_out/d/bench_default.json
for w in render c2 train-batch; do timeout 600 python bench.py --workload $w --no-cpu-baseline 2>>gpurun_out/d/bench.err | tail -1 > gpurun_out/d/bench_$w.json; done
python - <<'PY'
import json
for w in ("default","render","c2","train-batch"):
    d = json.load(open(f"gpurun_out/d/bench_{w}.json")); print(w, d["value"], d["detail"]["forward_ms"], d["detail"]["backward_ms"], d["detail"].get("value_repeated_frame"))
PY
