# PMC passes over the bench (counters only: no kernel-trace/stats mixing beyond --kernel-trace)
mkdir -p gpurun_out/pmc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_EXTRA"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
         "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/p$i -o run -- $BENCH > $R/gpurun_out/pmc/p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/pmc/p*/run_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
for k in agg:
    if 'rf::' not in k: continue
    print(k)
    for c in sorted(agg[k]): print('   %-28s %.4g (per launch, %d launches)'%(c, agg[k][c]/cnt[k][c], cnt[k][c]))
PY
