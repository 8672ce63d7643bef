# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
run() { # name lib extra...
  n=$1; lib=$2; shift 2
  RADFOAM_HIP_LIB=$GRAFT_REPO_ROOT/radfoam_amd/$lib timeout 400 python bench.py --workload train-batch --steps 6 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | tail -1 > $O/$n.json
  python - $n $O/$n.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); print(sys.argv[1], 'Mrays/s', d['value'], 'fwd', d['detail']['forward_ms'], 'bwd', d['detail']['backward_ms'])
except Exception as e: print(sys.argv[1], 'failed', e)
PY
}
for dens in "" "--empty-density 4.5e-6"; do
  tag=$([ -z "$dens" ] && echo sparse || echo alllit)
  run ${tag}_base libradfoam_hip.so $dens
  run ${tag}_dense_pitch libradfoam_hip.so $dens --grad-pitch dense
  run ${tag}_nomerge libradfoam_hip_nomerge.so $dens
  run ${tag}_staged libradfoam_hip_staged.so $dens
  run ${tag}_old libradfoam_hip_old.so $dens
  run ${tag}_old_dense_pitch libradfoam_hip_old.so $dens --grad-pitch dense
  run ${tag}_w4 libradfoam_hip_w4.so $dens
done
run alllit_sh2_base libradfoam_hip.so --empty-density 4.5e-6 --sh-degree 2
run alllit_sh2_old_dense libradfoam_hip_old.so --empty-density 4.5e-6 --sh-degree 2 --grad-pitch dense
