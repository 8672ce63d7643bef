# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4e; mkdir -p $O
cd $R
(timeout 900 python scripts/gpu_shard_sim.py 2>&1 | tail -30) > $O/shard_simulation_one_gpu.log; tail -6 $O/shard_simulation_one_gpu.log
cp gpurun_out/shard_sim*.json $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for m in static auto prev; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_tile_$m/$C -o run -- python $R/scripts/gpu_tile_order_asymmetric.py --mode $m > $O/pmc_tile_${m}_$C.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections, json, os, re
O = "gpurun_out/r4e"
out = {}
for m in ("static", "auto", "prev"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/pmc_tile_{m}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.search(r"rf::(\w+)", r["Kernel_Name"])
            if not k or "stats" in r["Kernel_Name"]:
                continue
            name = k.group(1)
            if name == "forward_kernel":
                name = "render" if re.search(r"forward_kernel<\d, true, true", r["Kernel_Name"]) else "forward"
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[m] = {k: {"launches": max(len(v) for v in c.values()),
                  "hbm_GB_per_launch": round((2 * sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1) +
                                              sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1)) * 1024 / 1e9, 3)}
              for k, c in agg.items() if k in ("forward", "render", "backward_replay_cached_kernel")}
json.dump(out, open(f"{O}/tile_order_asymmetric_hbm.json", "w"), indent=1)
print(json.dumps(out))
PY
