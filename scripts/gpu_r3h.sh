# Round 3, eighth GPU call: backward mode 5 (records sorted by cell and summed) against mode 4 on the flat batch.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3h
mkdir -p $O/ab
cd $R
(timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -80) > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
for extra in "" "--backward-mode 4" "--sh-degree 2" "--sh-degree 2 --backward-mode 4" "--quantiles 2"; do
  tag=$(echo "m5 $extra" | tr -d '-' | tr ' ' '_')
  (timeout 400 python bench.py --workload train-batch --steps 6 --warmup 2 --no-cpu-baseline $extra 2>$O/err_$tag.log | tail -1) > $O/ab/$tag.json
  python - "$tag" "$O/ab/$tag.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print(sys.argv[1], 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], 'fwd', d['detail']['forward_ms'], 'bwd', d['detail']['backward_ms'], 'pack', d['detail']['foam_pack_ms'])
except Exception as e:
    print(sys.argv[1], 'failed', e, open(sys.argv[2]).read()[-300:])
PY
done 2>&1 | tee $O/ab.log
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python $R/bench.py --workload train-batch --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -2) > $O/rocprof.log
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r3h/prof/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:12]
    for r in rows:
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e6, 3), "ms")
PY
