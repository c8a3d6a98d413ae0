# rocprofv3 summaries of the channeliser kernels (config 5's geometry): kernel-trace + stats, then HBM counters in their own passes.
#   gpurun --timeout 900 -- 'sh profiles/run_rocprof_chan.sh r05_a'
set -x
TAG=${1:-r05}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_chan
rm -rf $O && mkdir -p $O $GRAFT_REPO_ROOT/gpurun_out/prof_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python $GRAFT_REPO_ROOT/profiles/measure_chan.py > $O/trace.log 2>&1
f=$(find $O/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/prof_out/${TAG}_chan_kernel_stats.csv
grep '^{' $O/trace.log > $GRAFT_REPO_ROOT/gpurun_out/prof_out/${TAG}_chan.jsonl
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $GRAFT_REPO_ROOT/profiles/measure_chan.py > $O/pmc_$c.log 2>&1
    f=$(find $O/pmc_$c -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 - "$f" $c >> $GRAFT_REPO_ROOT/gpurun_out/prof_out/${TAG}_chan_pmc.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "channelise" in k:
        print(sys.argv[2], k, "dispatches", len(v), "mean", sum(v) / len(v))
PY
done
cd $GRAFT_REPO_ROOT && cat gpurun_out/prof_out/${TAG}_chan.jsonl && head -8 gpurun_out/prof_out/${TAG}_chan_kernel_stats.csv && cat gpurun_out/prof_out/${TAG}_chan_pmc.txt
