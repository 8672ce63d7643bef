# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -30) > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
(timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1) > $O/bench.json
(timeout 300 python scripts/gpu_tile_order_asymmetric.py 2>/dev/null | tail -1) > $O/tile_order_asymmetric_new_policy.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4f/bench.json")); r=d["roofline"]
print(d["value"], {k:r[k] for k in ("bound","frac","traffic","counters_stale","useful_valu_frac","useful_scan_valu_frac")}, r["isa_constants"])
for k,v in d["other_workloads"].items(): print(k, v.get("value"), v.get("matches_gpu_bitwise"), v.get("roofline",{}).get("bound"), v.get("roofline",{}).get("frac"), v.get("seconds"), v.get("error"))
t=json.load(open("gpurun_out/r4f/tile_order_asymmetric_new_policy.json"))
for m,r in t["result"].items(): print(m, {k:v["mean_ms"] for k,v in r.items() if isinstance(v,dict)})
PY
