# GPU: lower-MAC + receive-chain parity tests, decoder timing, chain leg of the bench under a kernel trace.
#   gpurun --timeout 1200 -- 'sh profiles/r06/run_lmac_check.sh <tag>'
TAG=${1:-r06x}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lmac.py tests/test_rx.py tests/test_burst_sync.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python profiles/measure_lmac.py > $O/measure_lmac.jsonl 2> $O/measure_lmac.err; cat $O/measure_lmac.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python $GRAFT_REPO_ROOT/bench.py --chain-only > $O/chain.json 2> $O/chain.err
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/chain_kernel_stats.csv
python - "$(find $O/trace -name '*kernel_trace.csv' | head -1)" > $O/trace_tail.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-26:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) if prev else 0
    print(f'{r["Kernel_Name"][:60]:60s} grid={r.get("Grid_Size_X","?"):>9s} dur_us={(e-s)/1e3:9.2f} gap_us={gap/1e3:8.2f}')
    prev = e
PY
rm -rf $O/trace
tail -9 $O/trace_tail.txt
python -c "
import json,sys
d=json.load(open('$O/chain.json'))
print(d['two_streams_ms_per_second'], d['one_stream_ms_per_second'], d['tail_ms_one_stream'], json.dumps({k:v['ms'] for k,v in d['stages_one_stream'].items()}))
print(sum(v['rows']-v['crc_good'] for v in d['check']['blocks'].values()), d['check']['channels_locked'], d['check']['cells_read'])
"
cd $GRAFT_REPO_ROOT; timeout 300 python profiles/measure_lmac_frames.py > $O/measure_lmac_frames.json 2> $O/measure_lmac_frames.err; cat $O/measure_lmac_frames.json; tail -3 $O/measure_lmac_frames.err
