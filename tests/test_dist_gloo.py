"""N > 1 path on CPU: two gloo ranks shard the channel axis, no data-path collective, results reassembled in
channel order and timing reduced with MAX -- the same host logic bench.py runs on RCCL."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_channel_range_partitions(pkg):
    cr = pkg.shard.channel_range
    for C in (1, 7, 16, 4096, 32768, 800):
        for W in (1, 2, 3, 4, 8):
            rs = [cr(C, W, r) for r in range(W)]
            assert rs[0][0] == 0 and rs[-1][1] == C
            assert all(rs[i][1] == rs[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    assert cr(32768, 8, 3) == (3 * 4096, 4 * 4096)
    with pytest.raises(ValueError):
        cr(16, 2, 2)


@pytest.mark.parametrize("C", [5, 8])
def test_two_rank_gloo_shards_and_reassembles(pkg, oracle, synth, tmp_path, C):
    N = 1500
    out = tmp_path / "r.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", "29613",
                    os.path.join(ROOT, "tests", "dist_worker.py"), str(out), str(C), str(N)],
                   check=True, env=env, timeout=300, cwd=ROOT)
    r = json.load(open(out))
    assert r["world"] == 2 and r["ranges"][0][0] == 0 and r["ranges"][1][1] == C
    assert r["tmax"] >= r["my_elapsed"]                      # MAX over ranks (rank 1 was made slower)
    rows = np.array(r["rows"])
    assert rows.shape[0] == C
    # single-process reference in channel order
    iq = np.stack([synth.gen_channel(N, 500 + c)[0] for c in range(C)])
    bits, nb, _, _ = oracle.process_batch(iq)
    assert np.array_equal(rows[:, 0], nb)
    assert np.array_equal(rows[:, 1:], bits[:, :64])
    # SUM over ranks (bench.py's per-rank known-answer counters): channels, bits, 0 + 1, one per rank
    assert r["sums"] == [C, int(nb.sum()), 1, 2]
