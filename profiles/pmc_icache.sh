# instruction-cache counters of k_fused (experiment): gpurun -- 'sh profiles/pmc_icache.sh 4096'
CH=${1:-4096}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_ic
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-check --no-host-path --no-large-batch --no-config5 --no-time-major --channels $CH"
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $O/ic -o p -- $B > $O/ic.log 2>&1
tail -2 $O/ic.log
python - <<PY
import csv,glob,collections
f=glob.glob("$O/ic/**/*counter_collection.csv", recursive=True)
acc=collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    if "k_fused" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print($CH, {k: sum(v[-6:])/6 for k,v in acc.items()})
PY
