# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m; mkdir -p $O
L=$GRAFT_REPO_ROOT/radfoam_amd
run() { n=$1; lib=$2; w=$3; shift 3
  RADFOAM_HIP_LIB=$L/$lib timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | tail -1 > $O/${n}_$w.json
  python -c "
import json; d=json.load(open('$O/${n}_$w.json')); print('$n', '$w', d['value'], d['detail']['forward_ms'], d['detail']['backward_ms'])"; }
for w in north-star train-batch; do
  run persistent16 libradfoam_hip_thr16.so $w --forward-mode 4
  run persistent0 libradfoam_hip_thr0.so $w --forward-mode 4
done
