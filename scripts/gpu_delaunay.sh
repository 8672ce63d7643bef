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
