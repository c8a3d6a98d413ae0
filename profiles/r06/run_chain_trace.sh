set -x
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06e}
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python $GRAFT_REPO_ROOT/bench.py --chain-only > $O/chain.json 2> $O/chain.err
f=$(find $O/trace -name '*kernel_stats.csv' | head -1)
cp $f $O/chain_kernel_stats.csv
t=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python - "$t" > $O/trace_tail.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 60 kernels: name, duration, gap
prev = None
for r in rows[-70:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) if prev else 0
    print(f'{r["Kernel_Name"][:60]:60s} grid={r.get("Grid_Size_X","?"):>9s} dur_us={(e-s)/1e3:9.2f} gap_us={gap/1e3:8.2f} q={r.get("Queue_Id","?")}')
    prev = e
PY
rm -rf $O/trace
head -30 $O/chain_kernel_stats.csv
