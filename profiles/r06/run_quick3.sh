TAG=${1:-r06x}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lmac.py tests/test_rx.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python profiles/measure_lmac_frames.py > $O/measure_lmac_frames.json 2>/dev/null; python -c "
import json
d=json.load(open('$O/measure_lmac_frames.json')); print(d['one_launch_long_first_ms'], d['one_launch_pool_scratch_ms'], d['launch_per_job_ms'], d['schf_ms_by_working_waves'])"
timeout 300 python profiles/measure_lmac.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin: d=json.loads(l); print(d['block'], d['ms'])"
for i in 1 2; do timeout 300 python bench.py --chain-only > $O/chain_$i.json 2>> $O/chain.err; python -c "
import json
d=json.load(open('$O/chain_$i.json')); print(d['two_streams_ms_per_second'], d['one_stream_ms_per_second'], d['tail_ms_one_stream'], {k:v['ms'] for k,v in d['stages_one_stream'].items()})"; done
