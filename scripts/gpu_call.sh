# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call: the training loop over 1000 and 3000 iterations on the final sources
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/cc
for n in 1000 3000; do
  timeout 400 python bench.py --workload train-loop --steps $n --no-cpu-baseline 2>gpurun_out/cc/loop$n.err | tail -1 > gpurun_out/cc/loop$n.json
  python - gpurun_out/cc/loop$n.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); det = d["detail"]
    print(d["value"], "it/s", det["ms_per_iteration"], det["rebuilds"])
except Exception as e:
    print("failed", e, open(sys.argv[1]).read()[-300:])
PY
done
