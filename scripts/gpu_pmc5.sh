# memory-pipeline counter passes (TA / TD / TCP) + SQ issue counters over the bench
mkdir -p gpurun_out/pmc5
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_EXTRA"
i=0
for C in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
         "TA_TA_BUSY_sum TA_BUSY_max TD_TD_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" ; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc5/p$i -o run -- $BENCH > $R/gpurun_out/pmc5/p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/pmc5/p*/run_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
for k in agg:
    if 'rf::forward' not in k and 'rf::backward_replay_c' not in k: continue
    print(k)
    for c in sorted(agg[k]): print('   %-36s %.4g'%(c, agg[k][c]/cnt[k][c]))
PY
tail -2 gpurun_out/pmc5/p*.log | grep -i "error\|invalid\|fail" | head
