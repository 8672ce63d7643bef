# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 f): shard simulations on the round-6 tree: the training batch with sorted shards, the north-star frame's rows
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/f
timeout 900 python scripts/gpu_shard_sim.py --batch --shards sorted > gpurun_out/f/shard_sim_training_batch_sorted.json 2>gpurun_out/f/sim.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/f/shard_sim_training_batch_sorted.json"))
print(d["replicated_per_rank_ms"], d["flat_grad_bytes"])
for w,r in d["worlds"].items(): print(w, r["slowest_rank_tracer_ms"], r["mean_rank_tracer_ms"], r["all_reduce_priced_ms"], r["step_ms_tracer_plus_exchange_plus_adam"], r.get("speedup_tracer_only"), r.get("speedup_step"), [ (x["forward_ms"], x["backward_ms"]) for x in r["ranks"][:2]])
PY
tail -3 gpurun_out/f/sim.err
timeout 900 python scripts/gpu_shard_sim.py --cuts balanced > gpurun_out/f/shard_sim_north_star.json 2>>gpurun_out/f/sim.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/f/shard_sim_north_star.json"))
for w,r in d["worlds"].items(): print(w, r["max_rank_device_ms_without_collectives"], r["mean_rank_device_ms_without_collectives"], r["scatter_all_ranks_ms"], r["zero_fill_flat_grad_ms"], [(x["pack_ms"], x["forward_ms"], x["backward_ms"], x["compact_ms"]) for x in r["ranks"][:3]])
PY
