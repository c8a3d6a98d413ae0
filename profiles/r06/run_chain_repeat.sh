# the chain leg a few times: run-to-run spread of the two-stream / one-stream figures
TAG=${1:-r06r}; N=${2:-4}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
for i in $(seq 1 $N); do
  timeout 300 python bench.py --chain-only > $O/chain_$i.json 2>> $O/chain.err
  python -c "
import json
d=json.load(open('$O/chain_$i.json')); print(d['two_streams_ms_per_second'], d['one_stream_ms_per_second'], d['tail_ms_one_stream'])"
done
