# Round 3, first GPU call: the new GPU tests (reference scene on the device, short last kd-blocks), the experimental
# owner-certified Delaunay build (tests + timing), forward scan variants A/B, L1/L2 counters of the training batch.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_reference_scene.py tests/test_delaunay.py tests/test_ref_binding.py -m gpu -q 2>&1 | tail -40) > $O/pytest_new.log; tail -3 $O/pytest_new.log
(RF_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_delaunay.py -m gpu -q -k owner 2>&1 | tail -40) > $O/pytest_owner.log; tail -3 $O/pytest_owner.log
for m in 0 1 2; do
  (RF_DELAUNAY_OWNER=$m RF_DELAUNAY_WAVES=owner$m timeout 300 python scripts/gpu_delaunay.py 2000000 5 2>&1 | grep -v amdgpu.ids | tail -3) > $O/delaunay_owner$m.log; tail -1 $O/delaunay_owner$m.log
done
for v in pipe2 pipe3; do
  (RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "forward_image_bit_exact or forward_flat_rays or benchmark_path or baseline_config_1 or large_image" 2>&1 | tail -5) > $O/pytest_$v.log; tail -1 $O/pytest_$v.log
done
VARIANTS="base pipe2 pipe3" AB_STEPS=10 bash scripts/gpu_ab.sh > $O/ab_north_star.log 2>&1; cat $O/ab_north_star.log
mkdir -p $O/ab_ns; cp gpurun_out/ab/*.json $O/ab_ns/
VARIANTS="base pipe2 pipe3" AB_STEPS=6 BENCH_EXTRA="--workload train-batch" bash scripts/gpu_ab.sh > $O/ab_train_batch.log 2>&1; cat $O/ab_train_batch.log
mkdir -p $O/ab_tb; cp gpurun_out/ab/*.json $O/ab_tb/
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --workload train-batch --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_ATOMIC_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TOTAL_ACCESSES_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_tb/p$i -o run -- $BENCH > $O/pmc_tb_p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json, os
out = {}
for f in sorted(glob.glob("gpurun_out/r3a/pmc_tb/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rf::" not in k: continue
        k = k.split("rf::")[1].split("<")[0].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(out, open("gpurun_out/r3a/train_batch_cache_counters.json", "w"), indent=1)
for k in out:
    if k.startswith(("forward", "backward")):
        print(k, {c: "%.4g" % v for c, v in out[k].items()})
# drop the raw csv of the other kernels (size)
for f in glob.glob("gpurun_out/r3a/pmc_tb/p*/**/*counter_collection.csv", recursive=True):
    rows = [l for j, l in enumerate(open(f)) if j == 0 or "rf::" in l]
    open(f, "w").writelines(rows)
PY
