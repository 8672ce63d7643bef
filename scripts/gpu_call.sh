# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 b): new GPU tests (certificate fail-safe, tile prior, sharded fetch), which test skips, the tile prior A/B
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/b
(timeout 900 python -m pytest tests -m gpu -q -x -rs 2>&1 | tail -12) > gpurun_out/b/pytest_gpu.log; tail -6 gpurun_out/b/pytest_gpu.log
timeout 600 python scripts/gpu_tile_prior.py --views 8 --res 32 48 > gpurun_out/b/tile_prior_north_star.json 2>gpurun_out/b/tp1.err; tail -c 3000 gpurun_out/b/tile_prior_north_star.json; tail -3 gpurun_out/b/tp1.err
timeout 600 python scripts/gpu_tile_prior.py --asymmetric --views 8 --res 32 > gpurun_out/b/tile_prior_asymmetric.json 2>gpurun_out/b/tp2.err; tail -c 3000 gpurun_out/b/tile_prior_asymmetric.json; tail -3 gpurun_out/b/tp2.err
