# the time-bounded GPU fuzzers on the round's final library (evidence files -> gpurun_out/<tag>/)
TAG=${1:-r06_fuzz}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1200 python profiles/fuzz_lmac_frames_gpu.py ${2:-600} $O/fuzz_lmac_frames_gpu.json 2> $O/fuzz_lmac.err | tail -1
timeout 700 python profiles/fuzz_parity.py --seconds ${3:-420} --seed 606 > $O/fuzz_parity.json 2> $O/fuzz_parity.err; python -c "
import json; d=json.load(open('$O/fuzz_parity.json')); print({k:(v if not isinstance(v,(list,dict)) else len(v)) for k,v in d.items()})"
timeout 500 python profiles/fuzz_bsync_gpu.py 606 ${4:-300} > $O/fuzz_bsync_gpu.json 2> $O/fuzz_bsync.err; tail -c 400 $O/fuzz_bsync_gpu.json
