# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call: SH-2 training batch, step counts and forward modes (the evidence run's 5.31 ms forward)
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
run2() {  # name steps warmup args...
  local name=$1 st=$2 wu=$3; shift 3
  timeout 300 python bench.py --steps $st --warmup $wu --no-other-workloads --no-cpu-baseline "$@" 2>gpurun_out/cc/$name.err | tail -1 > gpurun_out/cc/$name.json
  python - "$name" gpurun_out/cc/$name.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); det = d["detail"]
    print(sys.argv[1], "Mrays/s", d["value"], "fwd", det.get("forward_ms"), "bwd", det.get("backward_ms"))
except Exception as e:
    print(sys.argv[1], "failed", e, open(sys.argv[2]).read()[-300:])
PY
}
run2 sh2_auto_6   6 2  --workload train-batch --sh-degree 2
run2 sh2_auto_10  10 3 --workload train-batch --sh-degree 2
run2 sh2_m5_10    10 3 --workload train-batch --sh-degree 2 --forward-mode 5
run2 sh2_m2_10    10 3 --workload train-batch --sh-degree 2 --forward-mode 2
run2 sh2_auto_20  20 3 --workload train-batch --sh-degree 2
run2 sh2_auto_10b 10 3 --workload train-batch --sh-degree 2
run2 sh3_auto_20  20 3 --workload train-batch
run2 sh1_auto_10  10 3 --workload train-batch --sh-degree 1
run2 sh1_m2_10    10 3 --workload train-batch --sh-degree 1 --forward-mode 2
