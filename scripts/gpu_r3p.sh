# round 3, batch p: launch time of a row block against its height (strong-scaling staircase), base and hop-prefetch builds
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3p; cd $R
timeout 300 python scripts/gpu_shard_probe.py > gpurun_out/r3p/probe_base.jsonl 2> gpurun_out/r3p/probe_base.err
RADFOAM_HIP_LIB=$R/radfoam_amd/libradfoam_hip_hoppre.so timeout 300 python scripts/gpu_shard_probe.py > gpurun_out/r3p/probe_hoppre.jsonl 2> gpurun_out/r3p/probe_hoppre.err
VARIANTS="base hoppre" bash scripts/gpu_ab.sh > gpurun_out/r3p/ab_ns.log 2>&1
cat gpurun_out/r3p/ab_ns.log
paste -d'\n' gpurun_out/r3p/probe_base.jsonl gpurun_out/r3p/probe_hoppre.jsonl | cut -c1-150
