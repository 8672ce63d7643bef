# HBM traffic of the walk kernels: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (guide: TCC slots)
mkdir -p gpurun_out/traffic
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/traffic/fetch -o run -- $BENCH > $R/gpurun_out/traffic/fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/traffic/write -o run -- $BENCH > $R/gpurun_out/traffic/write.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, json
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/traffic/*/run_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
out={}
for k,v in agg.items():
    if 'rf::' not in k: continue
    f=sum(v.get('FETCH_SIZE',[0]))/max(len(v.get('FETCH_SIZE',[1])),1); w=sum(v.get('WRITE_SIZE',[0]))/max(len(v.get('WRITE_SIZE',[1])),1)
    # counters are in KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM):
    # report raw and the 2x-corrected upper estimate
    out[k]={'FETCH_SIZE_KB':f,'WRITE_SIZE_KB':w,'hbm_bytes_raw':(f+w)*1024,'hbm_bytes_fetch_x2':(2*f+w)*1024}
    print(k, out[k])
json.dump(out, open('gpurun_out/traffic/traffic.json','w'), indent=1)
PY
