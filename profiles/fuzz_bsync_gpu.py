#!/usr/bin/env python3
"""GPU-side fuzz of the burst synchroniser (round 5: four wavefronts per channel, frame-parallel LOCKED evaluation, packed frames):
k_burst_sync through the C ABI against the literal restatement fed one bit per call, on the adversarial streams of
tests/test_burst_sync.py (+ random bit flips), ragged per-channel calls incl. long ones (up to 36000 bits: ~70 frames per call, where the
frame-parallel path does most of the work), byte and packed frame outputs alternating.  python profiles/fuzz_bsync_gpu.py <seed> <seconds>.
Needs oracle/_ref (the burst builders).  Test infrastructure (uses oracle/ as the checker)."""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
import tetra_amd
from oracle import ref_binding as ref, binding as oracle
import test_burst_sync as T

pkg = tetra_amd.pkg
bb = pkg.bsync_binding
seed0, secs = int(sys.argv[1]), float(sys.argv[2])
rng = np.random.default_rng(seed0)
dev = torch.device("cuda", 0)
t0 = time.time()
rounds = channels = frames_total = batch_calls = 0
while time.time() - t0 < secs:
    Cn, max_bits = 64, 36000
    streams = []
    for c in range(Cn):
        tx = np.concatenate([T.make_stream(ref, int(rng.integers(1000, 10 ** 9))) for _ in range(int(rng.integers(1, 4)))])
        if rng.integers(0, 2):
            tx = tx ^ (rng.random(tx.size) < rng.uniform(0, 0.01)).astype(np.uint8)
        streams.append(tx)
    bs = bb.BurstSync(Cn, max_bits)
    F = bs.max_frames
    oracles = [oracle.BurstSyncOracle() for _ in range(Cn)]
    pos = np.zeros(Cn, np.int64)
    stride = (max_bits + 15) & ~15
    d_fr = torch.zeros((Cn, F, 512), dtype=torch.uint8, device=dev)
    d_fp = torch.zeros((Cn, F, 16), dtype=torch.int32, device=dev)
    d_ft = torch.zeros((Cn, F), dtype=torch.int32, device=dev)
    d_fb = torch.zeros((Cn, F), dtype=torch.int32, device=dev)
    d_nf = torch.zeros(Cn, dtype=torch.int32, device=dev)
    call = 0
    while (pos < np.array([s.size for s in streams])).any():
        rows = rng.integers(0, 2, (Cn, stride), dtype=np.uint8)
        nb = np.zeros(Cn, np.int32)
        for c in range(Cn):
            n = int(min(rng.choice([0, 1, 300, 4000, 9000, 20000, 36000]), streams[c].size - pos[c]))
            rows[c, :n] = streams[c][pos[c]:pos[c] + n]
            nb[c] = n
        packed = bool(call & 1)
        d_rows, d_nb = torch.from_numpy(rows).to(dev), torch.from_numpy(nb).to(dev)
        if packed:
            bs.process_packed_device(d_rows, stride, d_nb, d_fp, d_ft, d_fb, d_nf)
        else:
            bs.process_device(d_rows, stride, d_nb, d_fr, d_ft, d_fb, d_nf)
        torch.cuda.synchronize()
        ft, fb, nf = d_ft.cpu().numpy(), d_fb.cpu().numpy().view(np.uint32), d_nf.cpu().numpy()
        fr = d_fp.cpu().numpy().view(np.uint32) if packed else d_fr.cpu().numpy()
        st = bs.states()
        for c in range(Cn):
            fo = oracles[c].feed(streams[c][pos[c]:pos[c] + nb[c]], 1)
            pos[c] += nb[c]
            assert nf[c] == len(fo[0]), (seed0, rounds, call, c, "count")
            for k in range(nf[c]):
                if packed:
                    want = np.packbits(np.concatenate([fo[0][k], np.zeros(2, np.uint8)])).view(">u4").astype(np.uint32)
                    assert np.array_equal(fr[c, k], want), (seed0, rounds, call, c, k, "packed frame")
                else:
                    assert np.array_equal(fr[c, k, :510], fo[0][k]), (seed0, rounds, call, c, k, "frame")
            assert np.array_equal(ft[c, :nf[c]], fo[1]) and np.array_equal(fb[c, :nf[c]], fo[2]), (seed0, rounds, call, c, "types")
            assert st[c] == oracles[c].state, (seed0, rounds, call, c, "state")
            frames_total += int(nf[c])
            batch_calls += int(nf[c] >= 3)
        call += 1
    bs.close()
    rounds += 1
    channels += Cn
print(json.dumps(dict(rounds=rounds, channel_streams=channels, frames=frames_total, calls_with_3_or_more_frames=batch_calls,
                      seconds=round(time.time() - t0, 1), seed=seed0, failures=[])))
