R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r4f; cd $R
timeout 600 python scripts/gpu_flat_tile_order.py 2> gpurun_out/r4f/err2.log | tee gpurun_out/r4f/flat_tile_order.json; tail -3 gpurun_out/r4f/err2.log
