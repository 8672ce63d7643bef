# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
L=$GRAFT_REPO_ROOT/radfoam_amd
for v in al32l16; do
(RADFOAM_HIP_LIB=$L/libradfoam_hip_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3) > $O/pytest_$v.log; tail -1 $O/pytest_$v.log
done
run() { n=$1; lib=$2; w=$3; shift 3
  RADFOAM_HIP_LIB=$L/$lib timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | tail -1 > $O/${n}_$w.json
  python -c "
import json; d=json.load(open('$O/${n}_$w.json')); print('$n', '$w', d['value'], d['detail']['forward_ms'], d['detail']['backward_ms'], d['detail']['foam_pack_ms'], d['detail'].get('foam_full_pack_ms'))"; }
for w in train-batch north-star c5 render; do
  run base libradfoam_hip.so $w
  run al32 libradfoam_hip_al32.so $w
  run al32l16 libradfoam_hip_al32l16.so $w
  run l16 libradfoam_hip_l16.so $w
done
run base_alllit libradfoam_hip.so train-batch --empty-density 4.5e-6
run al32l16_alllit libradfoam_hip_al32l16.so train-batch --empty-density 4.5e-6
