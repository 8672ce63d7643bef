# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 zz): the bench lines again, now that counters.json describes these sources (counters_stale: false)
cd $GRAFT_REPO_ROOT
bash scripts/gpu_evidence.sh bench configs 2>&1 | grep -v "cpu:\|roofline\|loop:\|other:" | tail -22
