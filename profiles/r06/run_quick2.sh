# metric kernel alone (short bench) + chain leg twice
TAG=${1:-r06x}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --no-cpu-baseline --no-host-path --no-large-batch --no-config5 --no-time-major --no-chain > $O/bench_short.json 2> $O/bench_short.err; python -c "
import json
d=json.load(open('$O/bench_short.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['check'])"
for i in 1 2; do timeout 300 python bench.py --chain-only > $O/chain_$i.json 2>> $O/chain.err; python -c "
import json
d=json.load(open('$O/chain_$i.json')); print(d['two_streams_ms_per_second'], d['one_stream_ms_per_second'], d['tail_ms_one_stream'], {k:v['ms'] for k,v in d['stages_one_stream'].items()})"; done
