# Round 3, ninth GPU call: adaptive trail capacity (no re-walk launch for the few longest rays of a flat batch).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3i
mkdir -p $O/ab
cd $R
(timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60) > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
for extra in "" "--sh-degree 2" "--quantiles 2"; do
  tag=$(echo "tb $extra" | tr -d '-' | tr ' ' '_')
  (timeout 400 python bench.py --workload train-batch --steps 6 --warmup 3 --no-cpu-baseline $extra 2>$O/err_$tag.log | tail -1) > $O/ab/$tag.json
  python - "$tag" "$O/ab/$tag.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print(sys.argv[1], 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], 'fwd', d['detail']['forward_ms'], 'bwd', d['detail']['backward_ms'], 'pack', d['detail']['foam_pack_ms'])
except Exception as e:
    print(sys.argv[1], 'failed', e, open(sys.argv[2]).read()[-300:])
PY
done 2>&1 | tee $O/ab.log
(timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) > $O/ab/north_star.json
python -c "
import json; d=json.load(open('$O/ab/north_star.json')); print('north-star', d['value'], d['detail']['forward_ms'], d['detail']['backward_ms'])"
