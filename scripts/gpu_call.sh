# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 am): the triangulation's GPU tests eight times over and the 2 M foam twelve times (incremental + from scratch against Qhull): looking for anything intermittent in the shared-query second pass
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/am
for k in 1 2 3 4 5 6 7 8; do (timeout 600 python -m pytest tests/test_delaunay.py -m gpu -q -x 2>&1 | tail -1) >> gpurun_out/am/repeat.log; done
for k in 1 2 3 4; do timeout 300 python scripts/gpu_delaunay.py 2000000 5 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['equals_qhull'], d['incremental_equals_full'], d['stars_ms'], d['incremental_ms'])" >> gpurun_out/am/repeat.log; done
cat gpurun_out/am/repeat.log
