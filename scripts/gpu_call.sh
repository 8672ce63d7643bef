# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 g): the training batch with sorted shards; chunk:n group runs per XCD with every segment lit
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/g
timeout 900 python scripts/gpu_shard_sim.py --batch --shards sorted > gpurun_out/g/shard_sim_training_batch_sorted.json 2>gpurun_out/g/sim.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/g/shard_sim_training_batch_sorted.json"))
print(d["replicated_per_rank_ms"], d["flat_grad_bytes"])
for w,r in d["worlds"].items(): print(w, r["slowest_rank_tracer_ms"], r["mean_rank_tracer_ms"], r["all_reduce_priced_ms"], r["step_ms_tracer_plus_exchange_plus_adam"], r.get("speedup_tracer_only"), r.get("speedup_step"), [ (x["forward_ms"], x["backward_ms"]) for x in r["ranks"][:2]])
PY
tail -3 gpurun_out/g/sim.err
for t in default chunk:2 chunk:4 chunk:8 chunk:16 chunk:32; do
  X=""; [ "$t" != "default" ] && X="--tile-order $t"
  timeout 300 python bench.py --workload train-batch-lit --steps 10 --warmup 3 --no-cpu-baseline --no-repeated-frame $X 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$t', d['value'], d['detail']['forward_ms'], d['detail']['backward_ms'])"
done > gpurun_out/g/chunk_runs_all_lit.log 2>&1; cat gpurun_out/g/chunk_runs_all_lit.log
