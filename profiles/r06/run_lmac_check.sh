# GPU: lower-MAC + receive-chain parity tests, decoder timing, chain leg of the bench.  gpurun --timeout 1200 -- 'sh profiles/r06/run_lmac_check.sh <tag>'
TAG=${1:-r06x}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lmac.py tests/test_rx.py tests/test_burst_sync.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python profiles/measure_lmac.py > $O/measure_lmac.jsonl 2> $O/measure_lmac.err; cat $O/measure_lmac.jsonl
timeout 600 python bench.py --chain-only > $O/chain.json 2> $O/chain.err; python -c "
import json,sys
d=json.load(open('$O/chain.json'))
print(d['two_streams_ms_per_second'], d['one_stream_ms_per_second'], json.dumps(d['stages_one_stream']))
print(json.dumps(d['check']))
"
