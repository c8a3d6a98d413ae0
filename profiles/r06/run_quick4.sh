TAG=${1:-r06x}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rx.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python bench.py --chain-only > $O/chain.json 2>> $O/chain.err; python -c "
import json
d=json.load(open('$O/chain.json')); print(d['two_streams_ms_per_second'], d['one_stream_ms_per_second'], d['tail_ms_one_stream'], d['host_fetch_all_kinds_ms'], d['host_fetch_bytes'])"
