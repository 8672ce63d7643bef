# light SQ counter passes over the bench (two passes, 2 steps each)
mkdir -p gpurun_out/pmc_sq
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_EXTRA"
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_WAVES" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_sq/p$i -o run -- $BENCH > $R/gpurun_out/pmc_sq/p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/pmc_sq/p*/run_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
for k in agg:
    if 'rf::forward' not in k and 'rf::backward' not in k: continue
    print(k)
    for c in sorted(agg[k]): print('   %-36s %.4g'%(c, agg[k][c]/cnt[k][c]))
PY
