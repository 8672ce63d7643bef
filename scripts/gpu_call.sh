# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 w): GPU tests of the triangulation and of the reference's scene / loop on the final star code; the 500 k and 2 M foams timed
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/w
(timeout 900 python -m pytest tests/test_delaunay.py tests/test_reference_scene.py tests/test_gpu_fit.py -m gpu -q -x 2>&1 | tail -5) > gpurun_out/w/pytest.log; tail -3 gpurun_out/w/pytest.log
timeout 600 python scripts/gpu_delaunay.py 500000 1 2000000 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/w/delaunay.log
