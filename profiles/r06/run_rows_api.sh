TAG=${1:-r06x}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lmac.py tests/test_burst_sync.py tests/test_rx.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python profiles/measure_lmac.py 2>/dev/null | tee $O/measure_lmac.jsonl | python -c "
import json,sys
for l in sys.stdin: d=json.loads(l); print(d['block'], d['ms'])"
timeout 300 python profiles/measure_pipeline.py packed 2>/dev/null | tee $O/pipeline.jsonl | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if 'ms_lmac_x4' in d: print(d['second'], d['ms_burst_sync'], d['ms_demux_x4'], d['ms_lmac_x4'])"
