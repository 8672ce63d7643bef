# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call: ray order / hop trail keyed on the caller's strided view (collect_error_map): the new test, then the loop
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/cc
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_reference_scene.py -x -q -m gpu -k "strided or replay or reference or scene or loop" > gpurun_out/cc/pytest.log 2>&1; tail -3 gpurun_out/cc/pytest.log
timeout 400 python bench.py --workload train-loop --no-cpu-baseline 2>gpurun_out/cc/loop300.err | tail -1 > gpurun_out/cc/loop300.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/cc/loop300.json")); det = d["detail"]
print(d["value"], "it/s", det["ms_per_iteration"])
print({k: v for k, v in det["ms_per_call"].items() if "densif" in k or k in ("collect_error_map", "prune_and_densify")})
PY
