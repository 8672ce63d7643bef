# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "backward_parity or pitch or strict" 2>&1 | tail -3) > $O/pytest.log; tail -1 $O/pytest.log
run() { n=$1; lib=$2; shift 2
  RADFOAM_HIP_LIB=$GRAFT_REPO_ROOT/radfoam_amd/$lib timeout 400 python bench.py --workload train-batch --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads "$@" 2>/dev/null | tail -1 > $O/$n.json
  python -c "
import json; d=json.load(open('$O/$n.json')); print('$n', d['value'], d['detail']['forward_ms'], d['detail']['backward_ms'])"; }
for rep in 1 2; do
run pipe_sparse_$rep libradfoam_hip.so
run prev_sparse_$rep libradfoam_hip_prev.so
done
run pipe_alllit libradfoam_hip.so --empty-density 4.5e-6
run prev_alllit libradfoam_hip_prev.so --empty-density 4.5e-6
