mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
echo "=== bench 2M" > gpurun_out/bench.log
(timeout 600 python bench.py --steps 5 --warmup 1 --backward-mode 1 2>&1 | tail -3) >> gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o bench2m -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --backward-mode 1 2>&1 | tail -3) > $R/gpurun_out/rocprof.log
cd $R; ls -R gpurun_out/prof_r1 | head; cat gpurun_out/bench.log
