# chain leg of earlier commits' builds (.bisect/<commit>, built in the container) beside HEAD's
for d in "$@"; do
  (cd $GRAFT_REPO_ROOT/.bisect/$d && timeout 300 python bench.py --chain-only 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$d', d['two_streams_ms_per_second'], d['one_stream_ms_per_second'], d['tail_ms_one_stream'])")
done
cd $GRAFT_REPO_ROOT && timeout 300 python bench.py --chain-only 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('HEAD', d['two_streams_ms_per_second'], d['one_stream_ms_per_second'], d['tail_ms_one_stream'])"
