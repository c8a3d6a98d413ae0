# kernel trace of the two-stream steady state of the chain leg: a window of launches with start / end relative to the window's first
TAG=${1:-r06t}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o p -- python $GRAFT_REPO_ROOT/bench.py --chain-only > $O/chain.json 2> $O/chain.err
python - "$(find $O/trace -name '*kernel_trace.csv' | head -1)" > $O/overlap_window.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "anonymous" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the two-stream leg comes first: find the 9th..12th k_fused launches after the generator's kernels
fused = [i for i, r in enumerate(rows) if "k_fused" in r["Kernel_Name"]]
lo, hi = fused[8], fused[11]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi + 12]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0][:28]
    print(f'{name:28s} q={r.get("Queue_Id","?"):>3s} start_us={s/1e3:10.1f} end_us={e/1e3:10.1f} dur_us={(e-s)/1e3:9.1f}')
PY
rm -rf $O/trace
cat $O/overlap_window.txt
python -c "
import json
d=json.load(open('$O/chain.json')); print(d['two_streams_ms_per_second'], d['one_stream_ms_per_second'])"
