# Round 3, second GPU call: whole GPU suite (with tracebacks), the train-batch scheduling sweep, the default bench line
# with the new cpu_baseline.reference_source_envelope / self-verifying roofline.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -150) > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
(timeout 900 python scripts/gpu_train_batch_sweep.py --out $O/train_batch_sweep.json 2>&1 | grep -v amdgpu.ids | tail -40) > $O/sweep.log; cat $O/sweep.log
(timeout 900 python bench.py --steps 5 --warmup 2 2>$O/bench.err | tail -1) > $O/bench.json
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3b/bench.json").read())
    print(d["value"], d["roofline"]["frac"], d["roofline"].get("counters_stale"), d["roofline"].get("useful_valu_frac"), d["roofline"].get("fp32_frac_of_peak"))
    c = d["cpu_baseline"]
    print({k: v for k, v in c.items() if k not in ("sample", "reference_source_envelope")})
    e = c.get("reference_source_envelope")
    print("envelope:", e if not isinstance(e, dict) or "error" in e else {k: e[k] for k in ("seconds", "violations_of_the_bar")}, )
except Exception as ex:
    print("bench failed", ex, open("gpurun_out/r3b/bench.err").read()[-1500:])
PY
