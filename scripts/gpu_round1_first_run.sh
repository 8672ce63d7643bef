mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > gpurun_out/pytest_gpu.log
echo "=== bench 500k" > gpurun_out/bench.log
(timeout 400 python bench.py --points 500000 --seed 1 --steps 5 --warmup 1 --backward-mode 1 2>&1 | tail -5) >> gpurun_out/bench.log
echo "=== bench 2M" >> gpurun_out/bench.log
(timeout 600 python bench.py --steps 5 --warmup 1 --backward-mode 1 2>&1 | tail -5) >> gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1 -o bench2m -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --backward-mode 1 2>&1 | tail -5) > $R/gpurun_out/rocprof.log
cd $R; ls -R gpurun_out | head -50
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/bench.log
