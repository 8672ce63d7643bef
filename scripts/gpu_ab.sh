# A/B of kernel builds: VARIANTS="a b" -> radfoam_amd/libradfoam_hip_<v>.so (scripts/build_variant.sh), same bench line each
#   VARIANTS="base stage64" BENCH_EXTRA="--workload train-batch" bash scripts/gpu_ab.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ab
for v in $VARIANTS; do
  L=$R/radfoam_amd/libradfoam_hip_$v.so
  [ "$v" = "base" ] && L=$R/radfoam_amd/libradfoam_hip.so
  RADFOAM_HIP_LIB=$L timeout 400 python bench.py --steps ${AB_STEPS:-6} --warmup 2 --no-cpu-baseline --no-other-workloads $BENCH_EXTRA 2>/dev/null | tail -1 > $R/gpurun_out/ab/$v.json
  python - "$v" "$R/gpurun_out/ab/$v.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); w=d['detail'].get('walk',{})
    print(sys.argv[1], 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], 'fwd', d['detail']['forward_ms'], 'bwd', d['detail']['backward_ms'], 'pack', d['detail']['foam_pack_ms'],
          'lane util', round(w['cells_scanned']/max(w.get('wave_steps',1),1)/64,3) if w else '')
except Exception as e:
    print(sys.argv[1], 'failed', e, open(sys.argv[2]).read()[-300:])
PY
done
