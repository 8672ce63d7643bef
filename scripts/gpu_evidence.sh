# Evidence run of a round (on the GPU box, through gpurun): GPU parity suite, smoke, the bench line of every
# workload, the launcher paths, rocprofv3 kernel stats and the PMC passes behind bench.py's roofline.
#   bash scripts/gpu_evidence.sh [tests] [bench] [configs] [prof] [pmc]      (no argument = everything)
# Outputs under gpurun_out/ev/; scripts/update_profiles.py copies what is to be kept into profiles/.
# PMC passes: SQ+GRBM (two), FETCH_SIZE, WRITE_SIZE, L2 hit/miss, L1->L2 / L2->fabric request counts -- one --pmc run each.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev
mkdir -p $O
WHAT="${*:-tests bench configs prof pmc}"
has() { case " $WHAT " in *" $1 "*) return 0;; esac; return 1; }
cd $R
if has tests; then
  (timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25) > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
  (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > $O/smoke.log; tail -1 $O/smoke.log
fi
if has bench; then
  # the driver's line: north-star + the untimed other_workloads record (c2, c5, render, train-batch)
  (timeout 1500 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1) > $O/bench.json
  # the launcher paths with one GPU: bench.py's own (--gpus 1 needs none) and the driver's torch.distributed.run
  (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1) > $O/bench_torchrun1.json
fi
if has configs; then
  for w in c2 c5 train-batch train-batch-lit render; do
    (timeout 900 python bench.py --workload $w --steps 10 --warmup 3 2>$O/bench_$w.err | tail -1) > $O/bench_$w.json
  done
  (timeout 600 python bench.py --forward-only --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) > $O/bench_ns_fwd.json
  # the training loop's call pattern (train.py:176-180): two depth quantiles per ray with depth gradients
  (timeout 900 python bench.py --quantiles 2 --steps 10 --warmup 3 2>$O/bench_ns_q2.err | tail -1) > $O/bench_ns_q2.json
  (timeout 900 python bench.py --workload train-batch --quantiles 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) > $O/bench_train-batch_q2.json
  (timeout 900 python bench.py --workload train-batch --sh-degree 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) > $O/bench_train-batch_sh2.json
  # every segment lit (what the scene's softplus gives: real training), the reference's quotient scan, and the loop itself
  (timeout 900 python bench.py --strict-scan --steps 10 --warmup 3 2>$O/bench_ns_strict.err | tail -1) > $O/bench_ns_strict.json
  (timeout 900 python bench.py --workload train-loop 2>$O/bench_train-loop.err | tail -1) > $O/bench_train-loop.json
  [ -x scripts/probe/global_atomics ] && (timeout 120 scripts/probe/global_atomics > $O/probe_global_atomics.log 2>&1)
fi
cd /tmp && export TMPDIR=/tmp
PMCW="${PMC_WORKLOADS:-north-star c2 c5 render train-batch train-batch-lit}"
if has prof; then
  for w in $PMCW; do
    (timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o run -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-repeated-frame 2>&1 | tail -2) > $O/rocprof_$w.log
  done
fi
if has pmc; then
  python $R/radfoam_amd/build.py --source-hash > $O/csrc_sha256.txt   # the build these counters describe
  rocprofv3 -L > $O/counters_list.txt 2>&1
  for w in $PMCW; do
    BENCH="python $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-repeated-frame"
    # every workload: SQ+GRBM (the VALU-issue fraction), FETCH_SIZE, WRITE_SIZE (HBM bytes), L2 hit/miss; the workloads
    # named in PMC_FULL also the LDS / VMEM instruction mix and the L1->L2 / L2->fabric request counts
    case " ${PMC_FULL:-north-star train-batch train-batch-lit} " in *" $w "*) full=1;; *) full=0;; esac
    i=0
    for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" ; do
      i=$((i+1))
      if [ $full = 0 ] && { [ $i = 2 ] || [ $i = 6 ]; }; then continue; fi
      mkdir -p $O/pmc_$w
      timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$w/p$i -o run -- $BENCH > $O/pmc_$w/p$i.log 2>&1
    done
  done
fi
cd $R
python scripts/summarize_evidence.py gpurun_out/ev
