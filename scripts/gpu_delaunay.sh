# GPU triangulation: parity tests, timing on the BASELINE foams, per-kernel durations (gpurun)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then (timeout 600 python -m pytest tests/test_delaunay.py -m gpu -x -q 2>&1 | tail -15) > gpurun_out/pytest_delaunay.log; tail -3 gpurun_out/pytest_delaunay.log; fi
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_delaunay
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_delaunay -o run -- python $R/scripts/gpu_delaunay.py $DELAUNAY_ARGS > $R/gpurun_out/delaunay_run.log 2>&1; grep -v amdgpu.ids $R/gpurun_out/delaunay_run.log | tail -12
cd $R
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_delaunay/run_kernel_trace.csv")))
for r in rows:
    n = r["Kernel_Name"]
    if "delaunay_star" in n:
        print(n[:44], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2), "ms  grid", r.get("Grid_Size_X", r.get("Grid_Size")))
PY
if [ -n "$PMC" ]; then
  # counters of the two star kernels (separate passes, kernel trace only): instruction issue and HBM traffic
  cd /tmp
  i=0
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf $R/gpurun_out/pmc_delaunay/p$i; mkdir -p $R/gpurun_out/pmc_delaunay
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_delaunay/p$i -o run -- python $R/scripts/gpu_delaunay.py ${DELAUNAY_ARGS:-2000000 5} > $R/gpurun_out/pmc_delaunay/p$i.log 2>&1
  done
  cd $R
  python - <<PY
import csv, glob, collections
# the FIRST dispatch of each star kernel (a from-scratch build), counters summed over their per-XCD rows
for f in sorted(glob.glob("gpurun_out/pmc_delaunay/p*/run_counter_collection.csv")):
    first = {}; tot = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "delaunay_star" not in k: continue
        k = k.split("(")[0][-34:]
        d = int(r["Dispatch_Id"])
        first.setdefault(k, d)
        if d == first[k]: tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in tot:
        print(f.split("/")[2], k, {c: "%.4g" % v for c, v in tot[k].items()})
PY
fi
