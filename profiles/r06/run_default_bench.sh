# the driver's default line, timed, then the standard rocprof evidence of the metric kernel (profiles/run_rocprof.sh)
TAG=${1:-r06q}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
T0=$(date +%s.%N); python bench.py > $O/bench_default.json 2> $O/bench_default.err; T1=$(date +%s.%N); echo "default bench wall seconds: $(echo "$T1 - $T0" | bc)" | tee $O/time.txt
python -c "
import json
d=json.load(open('$O/bench_default.json'))
print({k:d[k] for k in ('metric','value','unit','ms_per_step','n_gpus','dtype','vs_baseline')})
print(d['roofline'])
print(d['cpu_baseline'])
print('config5', d['config5']['ms_per_step'], d['config5']['two_streams_ms_per_step'], d['config5']['config']['input_format'])
c=d['chain']; print('chain', c['two_streams_ms_per_second'], c['one_stream_ms_per_second'], c['tail_ms_one_stream'], {k:v['ms'] for k,v in c['stages_one_stream'].items()})
print(sorted(d.keys()))
"
sh profiles/run_rocprof.sh r06_q > $O/run_rocprof.log 2>&1; tail -12 $O/run_rocprof.log
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof      # raw rocprofv3 output (hundreds of MB): the summaries are in gpurun_out/prof_out
