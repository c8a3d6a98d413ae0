"""Training-sequence search against the REFERENCE ITSELF: oracle/_ref is the reference's
phy/tetra_burst.c compiled from /root/reference (oracle/build_ref.sh), so parity for this entry point is pinned."""
import numpy as np
import pytest

SEQS = None


def _seqs():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "etsi_training_sequences.json")) as f:
        ts = json.load(f)
    return {0: ts["normal_1"], 1: ts["normal_2"], 2: ts["normal_3"], 3: ts["sync"], 4: ts["extended"]}


def test_reference_library_finds_its_own_training_sequences(ref):
    seqs = _seqs()
    for t, s in seqs.items():
        row = np.zeros(600, np.uint8)
        row[100:100 + len(s)] = s
        assert ref.find_train_seq(row, 510) == (t, 100)
    rng = np.random.default_rng(0)
    nb = ref.build_norm_burst(rng.integers(0, 2, 216), rng.integers(0, 2, 30), rng.integers(0, 2, 216), 0)
    sb = ref.build_sync_burst(rng.integers(0, 2, 120), rng.integers(0, 2, 30), rng.integers(0, 2, 216))
    assert nb.size == 510 and sb.size == 510
    assert ref.find_train_seq(np.concatenate([nb, np.zeros(64, np.uint8)]), 510) == (ref.TRAIN_NORM_1, 244)
    assert ref.find_train_seq(np.concatenate([sb, np.zeros(64, np.uint8)]), 510) == (ref.TRAIN_SYNC, 214)


def test_reference_lookahead_quirk_exists(ref):
    """The reference's pre-filter skips in[20] for the first 21 positions: a sequence at position 3 is missed,
    the same sequence at position 30 is found (documented in include/tetra_burst_scan.h and reproduced on the GPU)."""
    s = _seqs()[0]
    row = np.zeros(200, np.uint8)
    row[3:3 + 22] = s
    early = ref.find_train_seq(row, 150)
    row2 = np.zeros(200, np.uint8)
    row2[30:30 + 22] = s
    assert ref.find_train_seq(row2, 150) == (0, 30)
    assert early != (0, 3)


@pytest.mark.gpu
def test_gpu_scan_equals_reference_on_planted_and_random_rows(pkg, ref):
    seqs = _seqs()
    rng = np.random.default_rng(42)
    Cn, stride = 600, 20000 + 64
    rows = rng.integers(0, 2, (Cn, stride), dtype=np.uint8)
    end = rng.integers(1, 20000, Cn).astype(np.int32)
    end[:40] = np.arange(1, 41)                       # tiny rows
    end[40:80] = 20000                                # full rows (multi-tile)
    for c in range(Cn):
        k = rng.integers(0, 4)
        for _ in range(k):
            t = int(rng.integers(0, 5))
            pos = int(rng.choice([rng.integers(0, 48), rng.integers(0, max(1, end[c])), max(0, end[c] - rng.integers(0, 45))]))
            s = seqs[t]
            if pos + len(s) <= stride:
                rows[c, pos:pos + len(s)] = s
    rows[100:130] = 0                                 # rows with nothing in them
    for mask in (0x1f, 0x08, 0x07, 0x10, 0x01):
        tg, og = pkg.scan_binding.find_train_seq_batch(rows, end, mask)
        for c in range(Cn):
            assert (int(tg[c]), int(og[c])) == ref.find_train_seq(rows[c], int(end[c]), mask), (c, mask, end[c])


@pytest.mark.gpu
def test_reference_bursts_through_the_gpu_demodulator(pkg, ref, synth):
    """Reference-built known answer for the whole path: continuous downlink bursts from the reference's own builders
    (tetra_burst.c:171-269) -> pi/4-DQPSK IQ -> GPU demodulator -> the reference's own tetra_find_train_seq locks onto
    the output at 510-bit slot spacing, and the GPU scan reports the same (type, offset) pairs."""
    rng = np.random.default_rng(7)
    Cn, nslots = 12, 40
    tx = []
    for c in range(Cn):
        slots = []
        for s in range(nslots):
            if s % 4 == 0:
                slots.append(ref.build_sync_burst(rng.integers(0, 2, 120), rng.integers(0, 2, 30), rng.integers(0, 2, 216)))
            else:
                slots.append(ref.build_norm_burst(rng.integers(0, 2, 216), rng.integers(0, 2, 30), rng.integers(0, 2, 216), s % 2))
        tx.append(np.concatenate(slots))
    N = nslots * 510 - 200
    iq = np.stack([synth.gen_channel(N, 300 + c, bits=tx[c])[0] for c in range(Cn)])
    d = pkg.Demodulator(Cn, N)
    bits, nb, _ = d.process(iq)
    d.close()
    # walk every channel with the reference finder, restarting after each hit, and with the GPU finder on the same rows
    for c in range(Cn):
        hits = []
        pos = 6000
        while pos < nb[c] - 600:
            t, o = ref.find_train_seq(bits[c][pos:], int(nb[c] - pos - 64))
            if t < 0:
                break
            hits.append((t, pos + o))
            pos += o + 60
        assert len(hits) >= 25, (c, hits[:5])
        sync_hits = [o for t, o in hits if t == ref.TRAIN_SYNC]
        assert len(sync_hits) >= 5 and all((b - a) % (4 * 510) == 0 for a, b in zip(sync_hits, sync_hits[1:]))
        norm_hits = [o for t, o in hits if t in (ref.TRAIN_NORM_1, ref.TRAIN_NORM_2)]
        assert all((b - a) % 510 == 0 for a, b in zip(norm_hits, norm_hits[1:]))
    # batched GPU scan of the tail of every channel == reference
    start = 8000
    sub = np.ascontiguousarray(bits[:, start:start + 4096 + 64])
    end = np.full(Cn, 4096, np.int32)
    tg, og = pkg.scan_binding.find_train_seq_batch(sub, end)
    for c in range(Cn):
        assert (int(tg[c]), int(og[c])) == ref.find_train_seq(sub[c], 4096), c


# ---------------------------------------------------------------------------------------------------------------------
# The plugin's own training-sequence indicator (src/main.cpp:385-414)
# ---------------------------------------------------------------------------------------------------------------------
IND_SEQS = {
    "n": [1,1, 0,1, 0,0, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,1, 0,0],
    "p": [0,1, 1,1, 1,0, 1,0, 0,1, 0,0, 0,0, 1,1, 0,1, 1,1, 1,0],
    "q": [1,0, 1,1, 0,1, 1,1, 0,0, 0,0, 0,1, 1,0, 1,0, 1,1, 0,1],
    "N": [1,1,1, 0,0,1, 1,0,1, 1,1,1, 0,0,0, 1,1,1, 1,0,0, 0,1,1, 1,1,0, 0,0,0, 0,0,0],
    "P": [1,0,1, 0,1,1, 1,1,1, 1,0,1, 0,1,0, 1,0,1, 1,1,0, 0,0,1, 1,0,0, 0,1,0, 0,1,0],
    "x": [1,0, 0,1, 1,1, 0,1, 0,0, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,1, 0,0, 0,0, 1,1],
    "X": [0,1,1,1,0,0,1,1,0,1,0,0,0,0,1,0,0,0,1,1,1,0,1,1,0,1,0,1,0,1,1,1,1,1,0,1,0,0,0,0,0,1,1,1,0],
    "y": [1,1, 0,0, 0,0, 0,1, 1,0, 0,1, 1,1, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,0, 0,0, 0,1, 1,0, 0,1, 1,1],
}


def test_indicator_restatement_arms_and_expires(oracle):
    """Known answers worked out by hand from main.cpp:385-414: a sequence of length L that ends at bit e is seen when the
    window's head holds it, 45 - L bits later; that bit arms 2048 and counts it down to 2047; 2047 further bits clear it."""
    for name, seq in IND_SEQS.items():
        L = len(seq)
        o = oracle.TsIndicatorOracle()
        pre = np.zeros(100, np.uint8)
        assert o.feed(pre) == (False, 0)
        assert o.feed(np.array(seq, np.uint8)) == ((True, 2047) if L == 45 else (False, 0)), name
        if L < 45:
            assert o.feed(np.zeros(45 - L - 1, np.uint8)) == (False, 0), name
            assert o.feed(np.zeros(1, np.uint8)) == (True, 2047), name
        again = 4 if name == "x" else 0        # the extended sequence holds normal sequence 1 at its bit 4: a second hit re-arms
        assert o.feed(np.zeros(2046, np.uint8)) == (True, 1 + again), name
        assert o.feed(np.zeros(again, np.uint8)) == (True, 1), name
        assert o.feed(np.zeros(1, np.uint8)) == (False, 0), name
    # the ETSI sequences of the fixture are the plugin's n / p / q / y / x
    ts = _seqs()
    assert ts[0] == IND_SEQS["n"] and ts[1] == IND_SEQS["p"] and ts[2] == IND_SEQS["q"] and ts[3] == IND_SEQS["y"] and ts[4] == IND_SEQS["x"]


@pytest.mark.gpu
def test_gpu_indicator_equals_the_restatement_chunked_with_carried_state(pkg, oracle):
    """600 channels of random bits with planted sequences (also across call boundaries and inside the first 44 bits of a
    call), ragged per-channel counts and call lengths from 0 to several tiles: tsfound and symsbeforeexpire after every call
    equal the literal restatement's; reset of one channel and of all."""
    rng = np.random.default_rng(7)
    Cn, total = 600, 60000
    rows = rng.integers(0, 2, (Cn, total), dtype=np.uint8)
    rows[:20] = 0                                           # quiet channels: only planted hits
    names = list(IND_SEQS)
    for c in range(Cn):
        for _ in range(int(rng.integers(0, 5))):
            s = IND_SEQS[names[int(rng.integers(0, 8))]]
            pos = int(rng.integers(0, total - 64))
            rows[c, pos:pos + len(s)] = s
    ind = pkg.scan_binding.TsIndicator(Cn)
    orcs = [oracle.TsIndicatorOracle() for _ in range(Cn)]
    pos = np.zeros(Cn, np.int64)
    for k, base_len in enumerate([1, 43, 44, 45, 0, 300, 2047, 2048, 2049, 9000, 17000, 8192, 31]):
        nb = np.minimum(base_len + (rng.integers(0, 40, Cn) if k % 2 else 0), total - pos).astype(np.int32)
        if base_len == 0:
            nb[:] = 0
        stride = (int(nb.max()) + 8 + 3) & ~3
        bits = np.zeros((Cn, stride), np.uint8)
        for c in range(Cn):
            bits[c, :nb[c]] = rows[c, pos[c]:pos[c] + nb[c]]
        found, expire = ind.process(bits, nb)
        for c in range(Cn):
            want = orcs[c].feed(rows[c, pos[c]:pos[c] + nb[c]])
            assert (bool(found[c]), int(expire[c])) == want, (k, c, base_len)
        pos += nb
        if k == 6:
            ind.reset(3)
            orcs[3] = oracle.TsIndicatorOracle()
    assert found[20:].any() and not found[20:].all()
    ind.reset()
    f, e = ind.process(np.zeros((Cn, 8), np.uint8), np.zeros(Cn, np.int32))
    assert not f.any() and not e.any()
    with pytest.raises(pkg.TetraDemodError):
        ind.reset(Cn)
    ind.close()


@pytest.mark.gpu
def test_gpu_indicator_sees_the_demodulated_downlink(pkg, synth):
    """End of the chain the plugin runs in NETSYMS mode: synthetic downlink bursts -> demodulator -> indicator: found on the
    burst channels, not on noise."""
    import torch
    Cn, N = 6, 36000
    iq = np.zeros((Cn, N), np.complex64)
    for c in range(Cn - 1):
        iq[c] = synth.gen_channel(N, 100 + c, bits=synth.gen_slot_bits(N // 510 + 2, 100 + c))[0]
    rng = np.random.default_rng(5)
    iq[Cn - 1] = (0.3 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))).astype(np.complex64)
    d = pkg.Demodulator(Cn, N)
    bits, nb, _ = d.process(iq)
    ind = pkg.scan_binding.TsIndicator(Cn)
    found, expire = ind.process(bits, nb)
    assert found[:Cn - 1].all() and not found[Cn - 1], (found, expire)
    ind.close()
    d.close()
