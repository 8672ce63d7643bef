# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call: block-level colour-row cache of the flat-batch replay (R rows, T table entries; rcC = occupancy control)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/cc
run() {  # name lib args...
  local name=$1 lib=$2; shift 2
  RADFOAM_HIP_LIB=$lib timeout 300 python bench.py --steps 6 --warmup 2 --no-other-workloads "$@" 2>gpurun_out/cc/$name.err | tail -1 > gpurun_out/cc/$name.json
  python - "$name" gpurun_out/cc/$name.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); det = d["detail"]
    cb = d.get("cpu_baseline") or {}
    print(sys.argv[1], "Mrays/s", d["value"], "fwd", det.get("forward_ms"), "bwd", det.get("backward_ms"),
          "bitwise", cb.get("matches_gpu_bitwise"), cb.get("points_grad_rel_l2"), cb.get("attr_grad_rel_l2"))
except Exception as e:
    print(sys.argv[1], "failed", e, open(sys.argv[2]).read()[-300:])
PY
}
B=$R/radfoam_amd/libradfoam_hip.so
for v in base rcA rcB rcC rcD rcE rcG; do
  L=$R/radfoam_amd/libradfoam_hip_$v.so; [ $v = base ] && L=$B
  run tb_$v $L --workload train-batch --no-cpu-baseline
  run lit_$v $L --workload train-batch --no-cpu-baseline --empty-density 4.5e-6
done
run tb_rcA_cpu $R/radfoam_amd/libradfoam_hip_rcA.so --workload train-batch --empty-density 4.5e-6
run tb_rcB_cpu $R/radfoam_amd/libradfoam_hip_rcB.so --workload train-batch --empty-density 4.5e-6
