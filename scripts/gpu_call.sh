# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l; mkdir -p $O
L=$GRAFT_REPO_ROOT/radfoam_amd
for v in base knn12 knn16 knn24; do
  lib=$L/libradfoam_hip_$v.so; [ $v = base ] && lib=$L/libradfoam_hip.so
  (RADFOAM_HIP_LIB=$lib timeout 400 python scripts/gpu_delaunay.py 500000 1 2000000 5 2>&1 | grep "^{" ) > $O/delaunay_$v.jsonl
  python - $v $O/delaunay_$v.jsonl <<'PY'
import json,sys
for l in open(sys.argv[2]):
    d=json.loads(l); print(sys.argv[1], d['points'], 'scratch', d['stars_ms'], 'inc', d['incremental_ms'], 'nodes', d['nodes_per_point'], 'ins', d['insertions_per_point'], 'qhull', d['equals_qhull'], d['incremental_equals_full'])
PY
done
