# final evidence of the round: whole GPU suite, smoke(), the driver's default bench line (timed)
TAG=${1:-r06_final}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
T0=$(date +%s); python bench.py > $O/bench_default.json 2> $O/bench_default.err; T1=$(date +%s); echo "default bench wall seconds: $((T1 - T0))" | tee $O/time.txt
python -c "
import json
d=json.load(open('$O/bench_default.json'))
print({k:d[k] for k in ('metric','value','unit','ms_per_step','n_gpus','dtype','vs_baseline')})
print('roofline', {k:d['roofline'][k] for k in ('bound','achieved','peak','frac','traffic')})
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
print('config5', d['config5']['ms_per_step'], d['config5']['two_streams_ms_per_step'])
c=d['chain']; print('chain', c['two_streams_ms_per_second'], c['one_stream_ms_per_second'], c['tail_ms_one_stream'], c['host_fetch_all_kinds_ms'], {k:v['ms'] for k,v in c['stages_one_stream'].items()})
"
