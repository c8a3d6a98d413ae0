"""Time-bounded fuzz of tetra_burst_index_device + tetra_lmac_decode_frames_device + tetra_lmac_track_sync_lists_device on the GPU
against (a) the decoder's lane code built for the host (tests/emul/lmac_emul.cpp, itself pinned on the reference's primitives by the
CPU suite) for every listed frame of every kind, and (b) the slot-layout tracker tetra_lmac_track_sync_device (pinned on the
reference's field read-out and TDMA arithmetic) for cell state / codes / times / SB1 labels.  Random frame types (incl. slots that
carry nothing), random payload with a share of reference-encoded blocks under the frame's own code, random codes, ragged frame
counts, carried cell state over several calls.  Usage: python profiles/fuzz_lmac_frames_gpu.py [seconds] [out.json]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tetra_amd  # noqa: E402
from tests.emul import lmac_emul_bind as E  # noqa: E402

pkg = tetra_amd.pkg
lb, bb = pkg.lmac_binding, pkg.bsync_binding
dev = torch.device("cuda", 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
KINDS = ((5, 0, 1, 288), (1, 2, 0, 144), (2, 1, 2, 144), (2, 2, 2, 144), (3, 0, 3, 32), (0, 1, 0, 80))       # tpsap, blk, list, type-2 row bytes
t_end = time.time() + budget
stats = dict(cases=0, blocks=0, crc_good=0, tracker_slots=0, differing=0)
seed = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    seed += 1
    C = int(rng.integers(1, 40))
    F = int(rng.choice([1, 7, 63, 64, 65, 72, 130]))
    n = C * F
    types = rng.choice(np.array([0, 1, 3, 3, 0, 1, 2, -1, -2], np.int32), n)
    frames = rng.integers(0, 2 ** 32, (n, 16), dtype=np.uint64).astype(np.uint32)
    codes = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    d_fr = torch.from_numpy(frames.astype(np.int64)).to(dev).to(torch.int32).contiguous()
    d_ft = torch.from_numpy(types).to(dev)
    d_codes = torch.from_numpy(codes.astype(np.int64)).to(dev).to(torch.int32)
    lists = torch.zeros((4, n), dtype=torch.int32, device=dev)
    counts = torch.zeros(4, dtype=torch.int32, device=dev)
    chan_first = torch.zeros((4, C), dtype=torch.int32, device=dev)
    bb.index_device(d_ft, F, lists, counts, chan_first)
    z = torch.zeros(n, dtype=torch.int32, device=dev)
    jobs, outs = [], []
    for tpsap, blk, li, n2 in KINDS:
        t2 = torch.zeros((n, n2), dtype=torch.uint8, device=dev)
        ok = torch.zeros(n, dtype=torch.int32, device=dev)
        outs.append((t2, ok))
        jobs.append(dict(type=tpsap, blk_num=blk, row_frame=lists[li], n_rows=counts[li:li + 1], max_rows=n, out_stride=n2,
                         frame_scramb=None if tpsap == 0 else d_codes, type2=t2, crc_ok=ok))
    lb.decode_frames_device(d_fr, d_ft, jobs, F, z, z, z)
    torch.cuda.synchronize()
    cnt = counts.cpu().numpy()
    hl = lists.cpu().numpy()
    want_lists = [np.flatnonzero(types == 3), np.flatnonzero(types == 0), np.flatnonzero(types == 1), np.flatnonzero(np.isin(types, (0, 1, 3)))]
    for k in range(4):
        if cnt[k] != want_lists[k].size or not np.array_equal(hl[k, :cnt[k]], want_lists[k]):
            stats["differing"] += 1
    for (tpsap, blk, li, n2), (t2, ok) in zip(KINDS, outs):
        rows = want_lists[li]
        if rows.size == 0:
            continue
        want, want_ok = E.decode_frames(tpsap, blk, frames, types, rows, None if tpsap == 0 else codes, n2)
        got, got_ok = t2[:rows.size].cpu().numpy(), ok[:rows.size].cpu().numpy()
        stats["differing"] += int((got != want).any(axis=1).sum() + (got_ok != want_ok).sum())
        stats["blocks"] += int(rows.size)
        stats["crc_good"] += int(want_ok.sum()) if tpsap != 3 else 0
    # the tracker on the decoded SB1 rows (random type-2 bits: every field value occurs), two calls with carried state
    sb1_t2, sb1_ok = outs[5]
    sync = want_lists[0]
    okc = torch.from_numpy((rng.random(n) < 0.7).astype(np.int32)).to(dev)         # most CRCs "good": the fields are random bits
    t2c = torch.from_numpy(rng.integers(0, 2, (n, 80), dtype=np.uint8)).to(dev)
    nf = torch.from_numpy(rng.integers(max(0, F - 5), F + 1, C).astype(np.int32)).to(dev)
    cell0 = torch.from_numpy(rng.integers(0, 70, (C, 10)).astype(np.int32)).to(dev)
    slot_t2 = torch.zeros((n, 80), dtype=torch.uint8, device=dev)
    slot_ok = torch.zeros(n, dtype=torch.int32, device=dev)
    slot_valid = torch.zeros(n, dtype=torch.int32, device=dev)
    if sync.size:
        idx = torch.from_numpy(sync).to(dev).long()
        slot_t2[idx], slot_ok[idx], slot_valid[idx] = t2c[:sync.size], okc[:sync.size], 1
    ca, cb = cell0.clone(), cell0.clone()
    oa = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3)]
    ob = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3)]
    for _ in range(2):
        lb.track_sync_device(slot_t2, 80, slot_ok, slot_valid, nf, C, F, ca, *oa)
        lb.track_sync_lists_device(t2c, 80, okc, d_ft, nf, chan_first[0], C, F, cb, *ob)
        torch.cuda.synchronize()
        stats["differing"] += int(not torch.equal(ca, cb)) + sum(int(not torch.equal(a, b)) for a, b in zip(oa, ob))
        stats["tracker_slots"] += n
    stats["cases"] += 1
stats["seconds"] = budget
print(json.dumps(stats))
if len(sys.argv) > 2:
    json.dump(stats, open(sys.argv[2], "w"))
if stats["differing"]:
    sys.exit(1)
