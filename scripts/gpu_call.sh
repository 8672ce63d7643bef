# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
L=$GRAFT_REPO_ROOT/radfoam_amd
for v in base td; do
  lib=$L/libradfoam_hip.so; [ $v = td ] && lib=$L/libradfoam_hip_td.so
  (RADFOAM_HIP_LIB=$lib timeout 300 python scripts/gpu_delaunay.py 2000000 5 2>&1 | tail -1) > $O/delaunay_$v.json; cat $O/delaunay_$v.json
done
(RADFOAM_HIP_LIB=$L/libradfoam_hip_refc.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "backward or frame or tile_order or trail or strict" 2>&1 | tail -5) > $O/pytest_refc.log; tail -2 $O/pytest_refc.log
for v in base base156 refc refc8 refc2; do
  lib=$L/libradfoam_hip_$v.so; [ $v = base ] && lib=$L/libradfoam_hip.so
  for w in north-star c2; do
    RADFOAM_HIP_LIB=$lib timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/${v}_$w.json
    python - $v $w $O/${v}_$w.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[3])); print(sys.argv[1], sys.argv[2], 'Mrays/s', d['value'], 'fwd', d['detail']['forward_ms'], 'bwd', d['detail']['backward_ms'])
except Exception as e: print(sys.argv[1], sys.argv[2], 'failed', e)
PY
  done
done
(timeout 600 python scripts/gpu_tile_order_asymmetric.py 2>&1 | tail -1) > $O/tile_order_asymmetric.json; cat $O/tile_order_asymmetric.json
