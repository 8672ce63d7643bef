# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
# this call (r06 z): the round's evidence on the final kernel sources -- scripts/gpu_evidence.sh (GPU suite, smoke, every bench line, kernel stats, PMC passes)
cd $GRAFT_REPO_ROOT
bash scripts/gpu_evidence.sh 2>&1 | tail -40
