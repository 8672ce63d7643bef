# Round 3, third GPU call: GPU suite; train-batch backward variants (point gradient through the block table = base,
# + in-register row merge, no global atomics at all = the floor of the replay); L2 hit/miss + atomic request counters.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -150) > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
VARIANTS="base merge noatomics" AB_STEPS=6 BENCH_EXTRA="--workload train-batch" bash scripts/gpu_ab.sh > $O/ab_train_batch.log 2>&1; cat $O/ab_train_batch.log
mkdir -p $O/ab_tb; cp gpurun_out/ab/*.json $O/ab_tb/
VARIANTS="base merge" AB_STEPS=6 BENCH_EXTRA="--workload train-batch --sh-degree 2" bash scripts/gpu_ab.sh > $O/ab_train_batch_sh2.log 2>&1; cat $O/ab_train_batch_sh2.log
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --workload train-batch --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_ATOMIC_sum" "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_tb/p$i -o run -- $BENCH > $O/pmc_tb_p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("gpurun_out/r3c/pmc_tb/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rf::" not in k: continue
        k = k.split("rf::")[1].split("<")[0].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = sum(v) / len(v)
    rows = [l for j, l in enumerate(open(f)) if j == 0 or "rf::" in l]
    open(f, "w").writelines(rows)
json.dump(out, open("gpurun_out/r3c/train_batch_cache_counters.json", "w"), indent=1)
for k in out:
    if k.startswith(("forward", "backward")):
        print(k, {c: "%.4g" % v for c, v in out[k].items()})
PY
