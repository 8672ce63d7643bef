mkdir -p gpurun_out/pmc2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_EXTRA"
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_LEVEL_WAVES" \
         "TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TD_TD_BUSY_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc2/p$i -o run -- $BENCH > $R/gpurun_out/pmc2/p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/pmc2/p*/run_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
for k in agg:
    if 'rf::forward' not in k and 'rf::backward' not in k: continue
    print(k)
    for c in sorted(agg[k]): print('   %-36s %.4g'%(c, agg[k][c]/cnt[k][c]))
PY
grep -il "error\|invalid" gpurun_out/pmc2/*.log | head
