# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 ag): the new GPU test of the second pass's other instantiations, and the triangulation's GPU tests
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_delaunay.py -m gpu -q -x 2>&1 | tail -4
