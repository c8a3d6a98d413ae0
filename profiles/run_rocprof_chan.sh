# rocprofv3 summaries of the channeliser kernels (config 5's geometry): kernel-trace + stats, then counters in their own passes
# (never combined with other trace domains).
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
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_out/${TAG}_chan_pmc.txt
n=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_SCA"; do
    n=$((n+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$n -o p -- python $GRAFT_REPO_ROOT/profiles/measure_chan_fft.py 0 > $O/pmc_$n.log 2>&1
    f=$(find $O/pmc_$n -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 - "$f" >> $GRAFT_REPO_ROOT/gpurun_out/prof_out/${TAG}_chan_pmc.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "channelise" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-28s %-36s dispatches %3d  mean %.1f" % (c, k, len(v), sum(v) / len(v)))
PY
done
cd $GRAFT_REPO_ROOT && cat gpurun_out/prof_out/${TAG}_chan.jsonl && head -5 gpurun_out/prof_out/${TAG}_chan_kernel_stats.csv && cat gpurun_out/prof_out/${TAG}_chan_pmc.txt
