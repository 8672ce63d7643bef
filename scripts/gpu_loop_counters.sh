# rocprofv3 kernel statistics + three PMC passes (SQ+GRBM, FETCH_SIZE, WRITE_SIZE) of the training loop's kernels
# (bench.py --workload train-loop), through gpurun; keeps only the rf:: rows of the counter files (the loop's torch kernels make
# them tens of MB) -> gpurun_out/loop/.   bash scripts/gpu_loop_counters.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/loop
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_loop -o run -- python $R/bench.py --workload train-loop --steps 120 > $O/rocprof_train-loop.log 2>&1
f=$(find /tmp/prof_loop -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f" > $O/train-loop_120_iterations_kernel_stats.csv
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_loop/p$i -o run -- python $R/bench.py --workload train-loop --steps 40 > /tmp/pmc_loop_p$i.log 2>&1
  g=$(find /tmp/pmc_loop/p$i -name "*counter_collection.csv" | head -1)
  mkdir -p $O/pmc_train-loop/p$i
  [ -n "$g" ] && (head -1 "$g"; grep "rf::" "$g") > $O/pmc_train-loop/p$i/run_counter_collection.csv
  rm -rf /tmp/pmc_loop/p$i
done
cd $R
python scripts/summarize_evidence.py gpurun_out/loop 2>&1 | grep "train-loop" | tee $O/summary.txt
du -sh $O
