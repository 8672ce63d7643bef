# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 e): contested cells resolved against the tournament's winner (scan_resolve) vs the dividing rescan; the LDS
# rows probe; the shard simulation of the data-parallel training step
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/e
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5) > gpurun_out/e/pytest_gpu.log; tail -3 gpurun_out/e/pytest_gpu.log
for w in north-star c2 render; do
  VARIANTS="base strictres base strictres" AB_STEPS=20 BENCH_EXTRA="--workload $w --no-repeated-frame" bash scripts/gpu_ab.sh 2>&1 | sed "s/^/$w /"
done > gpurun_out/e/resolve_ab.log; cat gpurun_out/e/resolve_ab.log
scripts/probe/lds_rows > gpurun_out/e/probe_lds_rows.log 2>&1; cat gpurun_out/e/probe_lds_rows.log
timeout 900 python scripts/gpu_shard_sim.py --batch > gpurun_out/e/shard_sim_training_batch.json 2>gpurun_out/e/sim.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/e/shard_sim_training_batch.json"))
print(d["replicated_per_rank_ms"], d["flat_grad_bytes"])
for w,r in d["worlds"].items(): print(w, r["slowest_rank_tracer_ms"], r["mean_rank_tracer_ms"], r["all_reduce_priced_ms"], r["step_ms_tracer_plus_exchange_plus_adam"], r.get("speedup_tracer_only"), r.get("speedup_step"), [ (x["forward_ms"], x["backward_ms"]) for x in r["ranks"][:2]])
PY
tail -3 gpurun_out/e/sim.err
