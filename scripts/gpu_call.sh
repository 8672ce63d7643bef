# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 af): rocprofv3 kernel trace + the four PMC passes of the triangulation's kernels on the final sources
cd $GRAFT_REPO_ROOT
SKIP_TESTS=1 PMC=1 DELAUNAY_ARGS="2000000 5" bash scripts/gpu_delaunay.sh 2>&1 | grep -v "^\[gpurun\]\|^W2026\|^E2026\|^I2026" | tail -30 | tee gpurun_out/delaunay_profile.log
