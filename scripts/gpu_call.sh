# the batch of one gpurun call (rewritten per call; what each call ran is recorded in profiles/README.md)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_delaunay
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_delaunay -o run -- python $R/scripts/gpu_delaunay.py 2000000 5 > $R/gpurun_out/delaunay_run.log 2>&1; grep "^{" $R/gpurun_out/delaunay_run.log | tail -2
cd $R
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/prof_delaunay/run_kernel_trace.csv")))
seq=[]
for r in rows:
    n = r["Kernel_Name"]
    if "rf::" in n or "rocprim" in n:
        seq.append((n.split("(")[0][-60:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r["Start_Timestamp"])))
# print the last incremental build's kernels: find the last 'delaunay_star_kernel' launches
idx=[i for i,(n,_,_) in enumerate(seq) if "delaunay_star_kernel" in n]
print(len(idx), "first-pass launches")
# launches: from-scratch x2 (reps), then moved: incremental x2, from scratch x1 -> incremental are idx[-3], idx[-2]
for k in (idx[-2],):
    t0=seq[k][2]
    for n,ms,ts in seq[k-8:k+14]:
        print("%-62s %9.3f ms  at %+9.3f ms" % (n, ms, (ts-t0)/1e6))
PY
