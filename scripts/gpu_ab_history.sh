# Re-measures two points of the kernel history on today's box, for the A/B the round-1 review asked to keep under
# profiles/: the LDS-staged walk (commit d1cca0b: the wave's distinct cells' face lists staged into LDS through a
# ballot/readlane dedupe, lanes scan from LDS) against its direct-scan successor (8e786df: lanes read the face table
# straight through L1).  Trees are exported by `git archive <commit>` into scripts/ab_trees/ (git-ignored) and built
# there; each runs ITS OWN bench.py on the same foam.   Output: gpurun_out/ab_history/<commit>.json + kernel stats.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_history
mkdir -p $O
for c in d1cca0b 8e786df; do
  T=$R/scripts/ab_trees/$c
  [ -d $T ] || { echo "missing $T"; continue; }
  rm -rf $T/.foam_cache; ln -s $R/.foam_cache $T/.foam_cache
  (cd $T && timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1) > $O/$c.json
  (cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -o run -- python $T/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
  python - $c $O/$c.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); print(sys.argv[1], d['value'], 'Mrays/s fwd', d['detail']['forward_ms'], 'bwd', d['detail']['backward_ms'])
except Exception as e: print(sys.argv[1], 'failed', e)
PY
  grep -h "forward_kernel\|backward" $O/prof_$c/run_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
done
