R=$GRAFT_REPO_ROOT
for v in $VARIANTS; do
  L=$R/radfoam_amd/libradfoam_hip_$v.so
  echo "== $v"
  RADFOAM_HIP_LIB=$L python scripts/gpu_bwdstats.py 2>/dev/null | tail -1
  RADFOAM_HIP_LIB=$L timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'],'Mrays/s fwd',d['detail']['forward_ms'],'bwd',d['detail']['backward_ms'])"
done
