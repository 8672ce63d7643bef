# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
L=$GRAFT_REPO_ROOT/radfoam_amd
run() { n=$1; lib=$2; shift 2
  RADFOAM_HIP_LIB=$L/$lib timeout 400 python bench.py --workload train-batch --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | tail -1 > $O/$n.json
  python -c "
import json; d=json.load(open('$O/$n.json')); print('$n', d['value'], d['detail']['forward_ms'], d['detail']['backward_ms'])"; }
for v in dt1024 dt1152; do
  run ${v}_sparse_b libradfoam_hip_$v.so
  run ${v}_alllit_b libradfoam_hip_$v.so --empty-density 4.5e-6
done
