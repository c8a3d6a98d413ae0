# rocprofv3 summaries of the resampler kernel (config 5's geometry): kernel-trace + stats, then counters in their own passes
# (never combined with other trace domains).
#   gpurun --timeout 900 -- 'sh profiles/run_rocprof_resamp.sh r06_a'
set -x
TAG=${1:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_resamp
rm -rf $O && mkdir -p $O $GRAFT_REPO_ROOT/gpurun_out/prof_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python $GRAFT_REPO_ROOT/profiles/measure_resamp.py > $O/trace.log 2>&1
f=$(find $O/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -14 $f > $GRAFT_REPO_ROOT/gpurun_out/prof_out/${TAG}_resamp_kernel_stats.csv
grep '^{' $O/trace.log > $GRAFT_REPO_ROOT/gpurun_out/prof_out/${TAG}_resamp.jsonl
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_out/${TAG}_resamp_pmc.txt
n=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
    n=$((n+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$n -o p -- python $GRAFT_REPO_ROOT/profiles/measure_resamp.py t16_w4 > $O/pmc_$n.log 2>&1
    f=$(find $O/pmc_$n -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 - "$f" >> $GRAFT_REPO_ROOT/gpurun_out/prof_out/${TAG}_resamp_pmc.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "resample" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-28s %-44s dispatches %3d  mean %.1f" % (c, k, len(v), sum(v) / len(v)))
PY
done
cd $GRAFT_REPO_ROOT && cat gpurun_out/prof_out/${TAG}_resamp.jsonl && head -8 gpurun_out/prof_out/${TAG}_resamp_kernel_stats.csv && cat gpurun_out/prof_out/${TAG}_resamp_pmc.txt
