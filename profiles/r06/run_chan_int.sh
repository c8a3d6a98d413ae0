# GPU: channeliser tests incl. integer input, config 5 with cs16 / cs8 / complex64 input, then the whole GPU suite.
TAG=${1:-r06x}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_chan.py tests/test_resamp.py -x -q -m gpu > $O/pytest_chan.log 2>&1; tail -3 $O/pytest_chan.log
for f in cs16 cs8 complex64; do
  timeout 300 python bench.py --config5 --config5-input $f > $O/config5_$f.json 2> $O/config5_$f.err
  python -c "
import json
d=json.load(open('$O/config5_$f.json'))
print('$f', d['ms_per_step'], d['two_streams_ms_per_step'], d['channeliser_kernel_ms'], d['resampler_kernel_ms'], d['demod_kernel_ms'], d['check'], d['roofline']['channeliser']['hbm'])
"
done
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; tail -3 $O/pytest_all.log
