"""Time-bounded fuzz of the receive chain's streaming logic (include/tetra_rx.h): a coded downlink per channel (random cell, random
lead-in of noise bits so that lock falls anywhere, Es/N0 14 .. 30 dB) cut into RANDOM calls -- two streams or one, every kind or a
random mask, results fetched one call late as a consumer would -- must deliver exactly the blocks (labels + type-1 bits) and the cell
states of ONE call over the whole stream.  What is compared is the handle against itself under another chunking: that the chain's
content is the reference's is the job of tests/test_rx.py.  Usage: python profiles/fuzz_rx_gpu.py [seconds] [out.json]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tetra_amd  # noqa: E402

pkg = tetra_amd.pkg
R, synth = pkg.rx_binding, pkg.synth
dev = torch.device("cuda", 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t_end = time.time() + budget
stats = dict(cases=0, calls=0, blocks=0, differing=0)


def collect(rx, kinds, which=0):
    out = {}
    for k in kinds:
        blocks, t1 = rx.fetch(k, which)
        out[k] = [(int(b["channel"]), int(b["bitnum"]), int(b["crc_ok"]), int(b["tdma_time_rx"]), int(b["tdma_time"]), t1[j].tobytes())
                  for j, b in enumerate(blocks)]
    return out


seed = 0
while time.time() < t_end:
    rng = np.random.default_rng(9000 + seed)
    seed += 1
    Cn = int(rng.integers(1, 7))
    nslots = int(rng.integers(12, 60))
    N = nslots * 510 - int(rng.integers(0, 400))
    iq = []
    for c in range(Cn):
        cell = (int(rng.integers(1, 1000)), int(rng.integers(1, 16000)), int(rng.integers(0, 64)))
        bits = synth.gen_downlink(nslots + 2, 77000 + 13 * seed + c, cell=cell)[0]
        lead = rng.integers(0, 2, 2 * int(rng.integers(0, 600)) + int(rng.integers(0, 2)) * 0).astype(np.uint8)      # (an even count: the modulator takes dibits)
        iq.append(synth.gen_channel(N, 88000 + 7 * seed + c, bits=np.concatenate([lead, bits])[: 2 * ((lead.size + bits.size) // 2)], esn0_db=float(rng.uniform(14, 30)))[0])
    iq = np.stack(iq)
    mask = 0 if rng.random() < 0.5 else int(rng.integers(1, 64))
    kinds = [k for k in range(R.N_KINDS) if mask == 0 or (mask >> k) & 1 or k == R.KIND_SB1]
    one = pkg.RxChain(Cn, N, kinds=mask)
    one.process(iq)
    one.wait()
    want = collect(one, kinds)
    want_cell = [tuple(getattr(c, f) for f, _ in R.CellState._fields_) for c in one.cells()]
    one.close()
    d_iq = torch.from_numpy(iq).to(dev)
    ncuts = int(rng.integers(1, 9))
    cuts = [0] + sorted(int(x) for x in rng.integers(1, N, ncuts)) + [N]
    cuts = sorted(set(cuts))
    maxlen = max(b - a for a, b in zip(cuts, cuts[1:]))
    flags = R.FLAG_ONE_STREAM if rng.random() < 0.4 else 0
    rx = pkg.RxChain(Cn, maxlen, flags=flags, kinds=mask)
    s = torch.cuda.Stream(dev)
    got = {k: [] for k in kinds}
    for i, (a, b) in enumerate(zip(cuts, cuts[1:])):
        chunk = d_iq[:, a:b].contiguous()
        s.wait_stream(torch.cuda.current_stream(dev))
        rx.process_device(chunk, b - a, s)
        chunk.record_stream(s)
        if i >= 1:
            prev = collect(rx, kinds, which=1)
            for k in kinds:
                got[k] += prev[k]
        stats["calls"] += 1
    last = collect(rx, kinds, which=0)
    for k in kinds:
        got[k] += last[k]
    rx.wait()
    cells = [tuple(getattr(c, f) for f, _ in R.CellState._fields_) for c in rx.cells()]
    rx.close()
    bad = sum(int(sorted(got[k]) != sorted(want[k])) for k in kinds) + int(cells != want_cell)
    stats["differing"] += bad
    stats["blocks"] += sum(len(want[k]) for k in kinds)
    stats["cases"] += 1
    if bad:
        stats.setdefault("first_failures", []).append(dict(seed=seed - 1, cuts=cuts, flags=flags, mask=mask))
stats["seconds"] = budget
print(json.dumps(stats))
if len(sys.argv) > 2:
    json.dump(stats, open(sys.argv[2], "w"))
sys.exit(1 if stats["differing"] else 0)
