"""Lower-MAC channel decoding against the REFERENCE ITSELF: oracle/_ref/libtetra_lmac_ref.so is
the reference's lower_mac/{tetra_scramb,tetra_interleave,tetra_conv_enc,crc_simple,viterbi,viterbi_cch,osmo_conv}.c compiled
from /root/reference (oracle/build_ref.sh), chained in the order tp_sap_udata_ind calls them (oracle/ref_binding.lmac_decode),
so parity for this entry point is pinned.  tests/golden/lmac_golden.npz holds inputs + the reference's outputs for boxes
without /root/reference or a prebuilt oracle/_ref."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CODED = (0, 1, 2, 4, 5)     # SB1, SB2, NDB, SCH/HU, SCH/F
STRIDE = 436                # a row stride that is not the row length


@pytest.fixture(scope="module")
def lref(ref):
    if not ref.lmac_available():
        pytest.skip("oracle/_ref/libtetra_lmac_ref.so not built and /root/reference not present")
    return ref


def make_rows(ref, blk_type, n, seed):
    """n input rows of one block kind: clean encoded blocks, encoded blocks with ~6 % bit errors, random bits, and
    arbitrary bytes (0xff / 0xfe / 2 / 7 ...: the reference's soft mapping has three classes)."""
    rng = np.random.default_rng(seed)
    n345, n2, n1, a, _ = ref.BLK_PARAM[blk_type]
    si = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    if blk_type == ref.TPSAP_T_SB1:
        si[:] = ref.SCRAMB_INIT
    rows = rng.integers(0, 256, (n, STRIDE), dtype=np.uint8)          # padding bytes are garbage on purpose
    sent = np.zeros((n, n1), np.uint8)
    for b in range(n):
        mode = b % 4
        if mode <= 1:
            sent[b] = rng.integers(0, 2, n1)
            r = ref.lmac_encode(blk_type, sent[b], si[b])
            if mode == 1:
                r = r ^ (rng.random(n345) < 0.06)
            rows[b, :n345] = r
        elif mode == 2:
            rows[b, :n345] = rng.integers(0, 2, n345)
        else:
            rows[b, :n345] = rng.choice(np.array([0, 1, 0xff, 0xfe, 2, 7], np.uint8), n345)
    return rows, si, sent


def ref_decode_rows(ref, blk_type, rows, si):
    n2 = ref.BLK_PARAM[blk_type][1]
    out = np.zeros((len(rows), n2), np.uint8)
    ok = np.zeros(len(rows), np.int32)
    for b in range(len(rows)):
        out[b], ok[b] = ref.lmac_decode(blk_type, rows[b], si[b])
    return out, ok


def test_reference_self_test_and_round_trip(lref):
    """The reference's own puncturer self-test (tetra_conv_enc.c:340, tetra_punct_test) passes on the library built here,
    and blocks made with its encoder primitives decode with crc_ok through its decoder primitives."""
    assert lref.lmac_lib().tetra_punct_test() == 0
    rng = np.random.default_rng(1)
    si = lref.scramb_get_init(262, 1, 5)
    for t in CODED:
        n345, n2, n1, a, _ = lref.BLK_PARAM[t]
        t1 = rng.integers(0, 2, n1).astype(np.uint8)
        t5 = lref.lmac_encode(t, t1, si)
        t2, ok = lref.lmac_decode(t, t5, si)
        assert ok == 1 and np.array_equal(t2[:n1], t1) and not t2[n1 + 16:].any()
        t5[rng.choice(n345, 3, replace=False)] ^= 1
        t2, ok = lref.lmac_decode(t, t5, si)
        assert ok == 1 and np.array_equal(t2[:n1], t1)


def test_lane_code_equals_reference(lref):
    """The kernel's lane-level source (csrc/lmac_core.hpp) built for the host == the reference, bit for bit, on clean,
    noisy, random and arbitrary-byte blocks of every coded kind."""
    from tests.emul import lmac_emul_bind
    for t in CODED:
        rows, si, sent = make_rows(lref, t, 240, 100 + t)
        want, want_ok = ref_decode_rows(lref, t, rows, si)
        got, got_ok = lmac_emul_bind.decode_batch(t, rows, si)
        assert np.array_equal(got, want), t
        assert np.array_equal(got_ok, want_ok), t
        n1 = lref.BLK_PARAM[t][2]
        clean = np.arange(len(rows)) % 4 == 0
        assert want_ok[clean].all() and np.array_equal(want[clean][:, :n1], sent[clean])
        assert 0 < want_ok.sum() < len(rows)


def test_lane_code_packed_route_equals_byte_route(lref):
    """Round 6: rows of plain bits (every byte 0 / 1) take a packed route through the decoder's front end (bytes -> bits, whole
    words of the scrambling sequence from a table that is linear in the code, bit spreading to classes); any other row the byte
    route (an LFSR step and a three-way classification per byte).  Same lane code as the kernel, on the host: the packed route IS
    taken for the clean rows, gives the byte route's output bit for bit, and both equal the reference."""
    import ctypes as C
    from tests.emul import lmac_emul_bind
    lmac_emul_bind.decode_batch(0, np.zeros((1, 120), np.uint8), np.zeros(1, np.uint32))          # builds / loads the library
    L = lmac_emul_bind._lib
    vp = C.c_void_p
    for t in CODED:
        n345, n2, n1, a, _ = lref.BLK_PARAM[t]
        rows, si, _ = make_rows(lref, t, 200, 300 + t)
        outs = []
        for route in (0, 1):
            out, ok, fast = np.zeros((len(rows), n2), np.uint8), np.zeros(len(rows), np.int32), C.c_int32(-1)
            assert L.lmac_emul_decode_route(n345, n2, n1, a, rows.ctypes.data_as(vp), len(rows), rows.shape[1], si.ctypes.data_as(vp),
                                            out.ctypes.data_as(vp), n2, ok.ctypes.data_as(vp), route, C.byref(fast)) == 0
            outs.append((out, ok, fast.value))
        clean = int(((rows[:, :n345] & 0xfe) == 0).all(1).sum())
        assert outs[0][2] == clean >= 150 and outs[1][2] == 0
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        want, want_ok = ref_decode_rows(lref, t, rows, si)
        assert np.array_equal(outs[0][0], want) and np.array_equal(outs[0][1], want_ok)


FRAME_KINDS = ((0, 1, 3), (1, 2, 3), (2, 1, 1), (2, 2, 1), (5, 0, 0), (3, 0, None))      # (tpsap, blk_num, carrying burst type)


def make_frames(lref, oracle, n, seed):
    """n packed frames of random burst types (NORM_1 / NORM_2 / SYNC, a few that carry nothing), each block position holding --
    in turn -- a block made with the reference's encoder (clean / with bit errors) under the frame's own scrambling code or random
    bits.  Returns (frames [n][512] bytes, packed [n][16], types, codes)."""
    from tests.emul import bsync_emul_bind as E
    rng = np.random.default_rng(seed)
    types = rng.choice(np.array([0, 1, 3, 3, 0, 1, 2, -1, -2], np.int32), n)
    codes = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    frames = np.zeros((n, 512), np.uint8)
    frames[:, :510] = rng.integers(0, 2, (n, 510))
    place = {(0, 1): [(94, 120)], (1, 2): [(282, 216)], (2, 1): [(14, 216)], (2, 2): [(282, 216)], (5, 0): [(14, 216), (282, 216)]}
    for r in range(n):
        for tpsap, blk, train in FRAME_KINDS[:5]:
            if types[r] != train or r % 3 == 2:
                continue
            n1 = lref.BLK_PARAM[tpsap][2]
            code = lref.SCRAMB_INIT if tpsap == 0 else int(codes[r])
            t5 = lref.lmac_encode(tpsap, rng.integers(0, 2, n1).astype(np.uint8), code)
            if r % 3 == 1:
                t5 = t5 ^ (rng.random(t5.size) < 0.05)
            at = 0
            for off, ln in place[(tpsap, blk)]:
                frames[r, off:off + ln] = t5[at:at + ln]
                at += ln
    return frames, E.pack_frames(frames), types, codes


def test_lane_code_from_frames_equals_demultiplexer_then_reference(lref, oracle):
    """tetra_lmac_decode_frames_device's lane code (round 6: the decoder's front end cuts its block out of the packed frame) on the
    host: for every kind, the frames that carry it -- and a few that do not (all-zero block) -- give exactly what the restated
    tetra_burst_rx_cb split (tetra_burst.c:343-393) followed by the REFERENCE's decoding chain gives."""
    from tests.emul import lmac_emul_bind
    frames, packed, types, codes = make_frames(lref, oracle, 400, 21)
    rng = np.random.default_rng(4)
    good_total = 0
    for tpsap, blk, train in FRAME_KINDS:
        carrying = [r for r in range(len(types)) if (types[r] == train if train is not None else types[r] in (0, 1, 3))]
        others = [r for r in range(len(types)) if r not in carrying][:7]
        listed = np.array(carrying + others, np.int32)
        rng.shuffle(listed)
        n2 = 32 if tpsap == 3 else lref.BLK_PARAM[tpsap][1]
        got, ok = lmac_emul_bind.decode_frames(tpsap, blk, packed, types, listed, None if tpsap == 0 else codes, n2 + 8)
        assert not got[:, n2:].any()
        for j, r in enumerate(listed):
            row = oracle.bsync_demux(frames[r], int(types[r]), tpsap, blk) if types[r] >= 0 else np.zeros(0, np.uint8)
            n345 = 30 if tpsap == 3 else lref.BLK_PARAM[tpsap][0]
            t5 = np.zeros(n345, np.uint8)
            t5[:row.size] = row
            code = lref.SCRAMB_INIT if tpsap == 0 else int(codes[r])
            want, want_ok = lref.lmac_decode(tpsap, t5, code)
            if tpsap == 3:
                want = np.concatenate([want, np.zeros(2, np.uint8)])          # the row's two padding bytes are written as zeros
            assert np.array_equal(got[j, :n2], want) and ok[j] == want_ok, (tpsap, blk, j, r)
            good_total += int(want_ok) if tpsap != 3 else 0
    assert good_total > 100
    with pytest.raises(ValueError):
        lmac_emul_bind.decode_frames(4, 0, packed, types, np.zeros(1, np.int32), codes, 112)      # SCH/HU: no downlink burst carries it


def test_lane_code_equals_golden():
    """Same check against the committed fixture (inputs + reference outputs; tests/golden/make_lmac_golden.py)."""
    from tests.emul import lmac_emul_bind
    g = np.load(os.path.join(HERE, "golden", "lmac_golden.npz"))
    for t in CODED:
        got, got_ok = lmac_emul_bind.decode_batch(t, g[f"rows_{t}"], g[f"scramb_{t}"])
        assert np.array_equal(got, g[f"type2_{t}"]) and np.array_equal(got_ok, g[f"crc_ok_{t}"])


@pytest.mark.gpu
def test_gpu_lmac_equals_golden(pkg):
    g = np.load(os.path.join(HERE, "golden", "lmac_golden.npz"))
    for t in CODED:
        n2 = g[f"type2_{t}"].shape[1]
        got, got_ok = pkg.lmac_binding.decode_batch(t, g[f"rows_{t}"], None if t == 0 else g[f"scramb_{t}"])
        assert np.array_equal(got[:, :n2], g[f"type2_{t}"]) and np.array_equal(got_ok, g[f"crc_ok_{t}"])
    got, got_ok = pkg.lmac_binding.decode_batch(3, g["rows_3"], g["scramb_3"])
    assert np.array_equal(got[:, :30], g["type2_3"]) and got_ok.all()


@pytest.mark.gpu
def test_gpu_lmac_equals_reference(pkg, lref):
    """Ragged batch sizes (1, 63, 64, 65, 1000 blocks), a padded and a tight row stride."""
    for t in CODED:
        n345, n2 = lref.BLK_PARAM[t][:2]
        rows, si, _ = make_rows(lref, t, 1000, 200 + t)
        want, want_ok = ref_decode_rows(lref, t, rows, si)
        for n in (1000, 65, 64, 63, 1):
            got, got_ok = pkg.lmac_binding.decode_batch(t, rows[:n], None if t == 0 else si[:n])
            assert np.array_equal(got[:, :n2], want[:n]), (t, n)
            assert np.array_equal(got_ok, want_ok[:n]), (t, n)
        tight = np.ascontiguousarray(rows[:200, :n345])
        got, got_ok = pkg.lmac_binding.decode_batch(t, tight, si[:200])
        assert np.array_equal(got[:, :n2], want[:200]) and np.array_equal(got_ok, want_ok[:200])


@pytest.mark.gpu
def test_gpu_lmac_packed_route_byte_route_and_mixed_workgroups(pkg, lref):
    """The kernel's two front ends: all-clean batches (packed route), the same batches with the byte route forced
    (tetra_lmac_debug_force_byte_route), batches where one row in every 64 holds an erasure byte (that workgroup falls back),
    a row stride that is not a multiple of 8 (byte route by alignment) -- all bit for bit the reference."""
    lb = pkg.lmac_binding
    rng = np.random.default_rng(77)
    for t in CODED:
        n345, n2, n1, a, _ = lref.BLK_PARAM[t]
        n = 300
        si = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        if t == lref.TPSAP_T_SB1:
            si[:] = lref.SCRAMB_INIT
        for stride in (n345 + (-n345) % 8, n345 + 4 if (n345 + 4) % 8 else n345 + 12, 1032):      # aligned (contiguous-run loader), 4-byte aligned and > 512 (byte route)
            rows = rng.integers(0, 2, (n, stride), dtype=np.uint8)
            for b in range(0, n, 3):
                rows[b, :n345] = lref.lmac_encode(t, rng.integers(0, 2, n1).astype(np.uint8), int(si[b])) ^ (rng.random(n345) < 0.03)
            want, want_ok = ref_decode_rows(lref, t, rows, si)
            got, ok = lb.decode_batch(t, rows, si)
            assert np.array_equal(got[:, :n2], want) and np.array_equal(ok, want_ok), (t, stride, "default")
            assert lb.force_byte_route(True) is False
            try:
                got, ok = lb.decode_batch(t, rows, si)
            finally:
                lb.force_byte_route(False)
            assert np.array_equal(got[:, :n2], want) and np.array_equal(ok, want_ok), (t, stride, "byte route")
            mixed = rows.copy()
            mixed[5::64, 7] = 0xff
            mixed[70, n345 - 1] = 2
            want, want_ok = ref_decode_rows(lref, t, mixed, si)
            got, ok = lb.decode_batch(t, mixed, si)
            assert np.array_equal(got[:, :n2], want) and np.array_equal(ok, want_ok), (t, stride, "mixed")


@pytest.mark.gpu
def test_gpu_lmac_bbk_and_argument_errors(pkg, lref):
    rng = np.random.default_rng(5)
    rows = rng.integers(0, 2, (300, 32), dtype=np.uint8)
    si = rng.integers(0, 2 ** 32, 300, dtype=np.uint64).astype(np.uint32)
    got, ok = pkg.lmac_binding.decode_batch(lref.TPSAP_T_BBK, rows, si)
    for b in range(300):
        t2, okr = lref.lmac_decode(lref.TPSAP_T_BBK, rows[b], si[b])
        assert np.array_equal(got[b, :30], t2) and ok[b] == okr == 1
    lb = pkg.lmac_binding
    with pytest.raises(pkg.TetraDemodError):                  # coded block without scrambling codes
        lb.decode_batch(lref.TPSAP_T_NDB, np.zeros((4, 216), np.uint8), None)
    with pytest.raises(pkg.TetraDemodError):                  # stride not a multiple of 4
        lb.decode_batch(lref.TPSAP_T_SB1, np.zeros((4, 121), np.uint8), None)
    with pytest.raises(pkg.TetraDemodError):                  # row shorter than the block
        lb.decode_batch(lref.TPSAP_T_SCH_F, np.zeros((4, 216), np.uint8), np.zeros(4, np.uint32))
    with pytest.raises(pkg.TetraDemodError):
        lb.decode_batch(7, np.zeros((4, 216), np.uint8), np.zeros(4, np.uint32))
    out, ok = lb.decode_batch(lref.TPSAP_T_SB1, np.zeros((0, 120), np.uint8), None)   # empty batch is a no-op
    assert out.shape[0] == 0 and ok.size == 0


@pytest.mark.gpu
def test_gpu_iq_to_sync_pdu(pkg, lref, synth):
    """Whole receive path on the GPU with a reference-built transmit side: SYNC PDUs -> reference encoder primitives ->
    reference burst builder -> pi/4-DQPSK IQ -> GPU demodulator -> GPU training-sequence search -> block extraction at the
    reference's SB_BLK1_OFFSET -> GPU lower-MAC decode -> CRC good and the PDU bits recovered."""
    rng = np.random.default_rng(11)
    Cn, nslots = 8, 36
    SB_BLK1_OFFSET, SYNC_TRAIN_OFFSET = 94, 214         # tetra_burst.c:33 and the sync training sequence position
    pdus, tx = [], []
    for c in range(Cn):
        slots = []
        for s in range(nslots):
            if s % 3 == 0:
                t1 = rng.integers(0, 2, 60).astype(np.uint8)
                pdus.append((c, t1))
                slots.append(lref.build_sync_burst(lref.lmac_encode(lref.TPSAP_T_SB1, t1, 3), rng.integers(0, 2, 30), rng.integers(0, 2, 216)))
            else:
                slots.append(lref.build_norm_burst(rng.integers(0, 2, 216), rng.integers(0, 2, 30), rng.integers(0, 2, 216), s % 2))
        tx.append(np.concatenate(slots))
    N = nslots * 510 - 200
    iq = np.stack([synth.gen_channel(N, 900 + c, bits=tx[c])[0] for c in range(Cn)])
    d = pkg.Demodulator(Cn, N)
    bits, nb, _ = d.process(iq)
    d.close()
    # sync-burst search window by window (3 slots = 1530 bits per window, after the loops have locked)
    rows, owner = [], []
    for c in range(Cn):
        pos = 7000
        while pos + 1530 + 600 < nb[c]:
            win = np.ascontiguousarray(bits[c:c + 1, pos:pos + 1530 + 66])   # stride % 4 == 0
            t, o = pkg.scan_binding.find_train_seq_batch(win, np.array([1530], np.int32), 1 << lref.TRAIN_SYNC)
            if t[0] != lref.TRAIN_SYNC:
                pos += 1530
                continue
            burst = pos + int(o[0]) - SYNC_TRAIN_OFFSET
            if burst >= 0:
                rows.append(bits[c, burst + SB_BLK1_OFFSET: burst + SB_BLK1_OFFSET + 120])
                owner.append(c)
            pos += int(o[0]) + 60
    assert len(rows) >= Cn * 4
    type2, ok = pkg.lmac_binding.decode_batch(lref.TPSAP_T_SB1, np.stack(rows), None)
    for r in range(len(rows)):                          # bit-exact with the reference on the demodulated rows
        t2, okr = lref.lmac_decode(lref.TPSAP_T_SB1, rows[r], 3)
        assert np.array_equal(type2[r, :80], t2) and ok[r] == okr
    assert ok.mean() > 0.9                              # a channel may still be settling at the first window
    sent = {c: [p.tobytes() for cc, p in pdus if cc == c] for c in range(Cn)}
    for r in range(len(rows)):
        if ok[r]:
            assert type2[r, :60].tobytes() in sent[owner[r]]


@pytest.mark.gpu
def test_gpu_lmac_device_entry_with_wide_output_rows_on_a_side_stream(pkg, lref):
    """The *_device entry point on a non-default stream, output rows wider than type2_bits (the padding must be left
    untouched), two back-to-back launches sharing the stream-ordered decision scratch."""
    import torch
    lb = pkg.lmac_binding
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    t = lref.TPSAP_T_SCH_F
    rows, si, _ = make_rows(lref, t, 333, 909)
    want, want_ok = ref_decode_rows(lref, t, rows, si)
    d_rows = torch.from_numpy(rows).to(dev)
    d_si = torch.from_numpy(si.view(np.int32)).to(dev)
    outs = []
    torch.cuda.synchronize()
    for rep in range(2):
        d_out = torch.full((333, 304), 7, dtype=torch.uint8, device=dev)
        d_ok = torch.full((333,), -1, dtype=torch.int32, device=dev)
        lb.decode_batch_device(t, d_rows, 333, STRIDE, d_si, d_out, 304, d_ok, st)
        outs.append((d_out, d_ok))
    st.synchronize()
    for d_out, d_ok in outs:
        out, ok = d_out.cpu().numpy(), d_ok.cpu().numpy()
        assert np.array_equal(out[:, :288], want) and (out[:, 288:] == 7).all() and np.array_equal(ok, want_ok)


@pytest.mark.gpu
def test_gpu_lmac_full_size_round_trip(pkg, lref):
    """One second of 4096 channels (286 720 SCH/F blocks in one call): encode -> flip up to 4 of the 432 bits -> decode
    gives the payload back with a good CRC for (almost) every block, a good CRC always means the sent payload
    (size-independent properties), and the uncorrectable blocks plus a random sample equal the reference decode bit for bit."""
    rng = np.random.default_rng(31)
    t = lref.TPSAP_T_SCH_F
    n_distinct, n = 1024, 4096 * 70
    si_d = rng.integers(0, 2 ** 32, n_distinct, dtype=np.uint64).astype(np.uint32)
    pay = rng.integers(0, 2, (n_distinct, 268), dtype=np.uint8)
    enc = np.stack([lref.lmac_encode(t, pay[k], si_d[k]) for k in range(n_distinct)])
    pick = rng.integers(0, n_distinct, n)
    rows = enc[pick]
    flips = rng.integers(0, 432, (n, 4))
    nflip = rng.integers(0, 5, n)
    for j in range(4):
        m = nflip > j
        rows[np.nonzero(m)[0], flips[m, j]] ^= 1
    out, ok = pkg.lmac_binding.decode_batch(t, rows, si_d[pick])
    good = ok == 1
    assert good.mean() > 0.97                                    # <= 4 channel errors are usually corrected (98.9 % here)
    assert np.array_equal(out[good][:, :268], pay[pick][good])   # and a good CRC means the payload is back
    check = np.concatenate([np.nonzero(~good)[0][:300], rng.integers(0, n, 200)])
    for r in check:                                              # uncorrectable blocks and a random sample: == reference
        t2, okr = lref.lmac_decode(t, rows[r], si_d[pick[r]])
        assert np.array_equal(out[r, :288], t2) and okr == ok[r]


@pytest.mark.gpu
def test_gpu_track_scramb_equals_reference_rule(pkg, lref):
    """tetra_lmac_track_scramb_device == the reference's rule (tetra_lower_mac.c:258-266 with tetra_scramb_get_init from the
    reference build): per channel in time order, a valid SB1 row with a good CRC replaces the code; carried across calls."""
    import torch
    rng = np.random.default_rng(8)
    Cn, F = 37, 23
    dev = torch.device("cuda", 0)
    chan = rng.integers(0, 2 ** 32, Cn, dtype=np.uint64).astype(np.uint32)
    chan[:5] = 0
    want_chan = chan.copy()
    d_chan = torch.from_numpy(chan.view(np.int32).copy()).to(dev)
    for call in range(3):
        t2 = rng.integers(0, 2, (Cn * F, 80), dtype=np.uint8)
        ok = (rng.random(Cn * F) < 0.3).astype(np.int32)
        valid = (rng.random(Cn * F) < 0.4).astype(np.int32)
        want_rows = np.zeros(Cn * F, np.uint32)
        for c in range(Cn):
            cur = want_chan[c]
            for f in range(F):
                r = c * F + f
                if valid[r] and ok[r]:
                    bits = t2[r]
                    val = lambda a, n: int("".join(map(str, bits[a:a + n])), 2)
                    cur = np.uint32(lref.scramb_get_init(val(31, 10), val(41, 14), val(4, 6)))
                want_rows[r] = cur
            want_chan[c] = cur
        d_rows = torch.zeros(Cn * F, dtype=torch.int32, device=dev)
        pkg.lmac_binding.track_scramb_device(torch.from_numpy(t2).to(dev), 80, torch.from_numpy(ok).to(dev), torch.from_numpy(valid).to(dev),
                                             Cn, F, d_chan, d_rows)
        torch.cuda.synchronize()
        assert np.array_equal(d_rows.cpu().numpy().view(np.uint32), want_rows), call
        assert np.array_equal(d_chan.cpu().numpy().view(np.uint32), want_chan), call


class _TdmaTime(__import__("ctypes").Structure):
    """struct tetra_tdma_time (src/decoder/src/tetra_tdma.h:6-12)"""
    import ctypes as _C
    _fields_ = [("hn", _C.c_uint16), ("sn", _C.c_uint32), ("tn", _C.c_uint32), ("fn", _C.c_uint32), ("mn", _C.c_uint32)]


@pytest.mark.gpu
def test_gpu_track_sync_equals_the_reference_rule_and_clock(pkg, lref, ref):
    """tetra_lmac_track_sync_device == tp_sap_udata_ind's SB1 case (tetra_lower_mac.c:246-275) + the LOCKED receiver's clock
    (tetra_burst_sync.c:113), walked per channel in frame order with the REFERENCE'S OWN tetra_tdma_time_add_tn
    (oracle/_ref: src/decoder/src/tetra_tdma.c compiled where it lies) and tetra_scramb_get_init: a SYNC PDU with a good CRC
    sets colour code / TN / FN / MN / MCC / MNC and the scrambling code; the PHY time takes tcd's after EVERY SB1, good CRC or
    not; every consumed frame advances it by one timeslot with the reference's wrap thresholds.  Three calls, state carried,
    ragged frame counts; arbitrary field values (FN up to 31, MN up to 63 as the bit fields allow) exercise the normalisation."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(9)
    Cn, F = 41, 29
    dev = torch.device("cuda", 0)
    add_tn = ref.lib().tetra_tdma_time_add_tn
    add_tn.argtypes = [C.POINTER(_TdmaTime), C.c_uint32]
    add_tn.restype = None
    cell = np.zeros((Cn, 10), np.uint32)
    d_cell = torch.from_numpy(cell.view(np.int32).copy()).to(dev)
    want = [dict(scr=0, cc=0, mcc=0, mnc=0, tcd=(0, 0, 0), phy=_TdmaTime()) for _ in range(Cn)]
    for call in range(3):
        t2 = rng.integers(0, 2, (Cn * F, 80), dtype=np.uint8)
        ok = (rng.random(Cn * F) < 0.5).astype(np.int32)
        valid = (rng.random(Cn * F) < 0.3).astype(np.int32)
        nfr = rng.integers(0, F + 1, Cn).astype(np.int32)
        w_scr = np.zeros(Cn * F, np.uint32)
        w_rx = np.zeros(Cn * F, np.uint32)
        w_t = np.zeros(Cn * F, np.uint32)
        for c in range(Cn):
            w = want[c]
            for f in range(F):
                r = c * F + f
                if f < nfr[c]:
                    add_tn(C.byref(w["phy"]), 1)
                    w_rx[r] = w["phy"].tn | (w["phy"].fn << 8) | (w["phy"].mn << 16)
                    if valid[r]:
                        bits = t2[r]
                        val = lambda a, n: int("".join(map(str, bits[a:a + n])), 2)
                        if ok[r]:
                            w["cc"], w["mcc"], w["mnc"] = val(4, 6), val(31, 10), val(41, 14)
                            w["tcd"] = (val(10, 2) + 1, val(12, 5), val(17, 6))
                            w["scr"] = int(lref.scramb_get_init(w["mcc"], w["mnc"], w["cc"]))
                        w["phy"].tn, w["phy"].fn, w["phy"].mn = w["tcd"]
                    w_t[r] = w["phy"].tn | (w["phy"].fn << 8) | (w["phy"].mn << 16)
                w_scr[r] = w["scr"]
        d_scr = torch.zeros(Cn * F, dtype=torch.int32, device=dev)
        d_rx = torch.zeros(Cn * F, dtype=torch.int32, device=dev)
        d_t = torch.zeros(Cn * F, dtype=torch.int32, device=dev)
        pkg.lmac_binding.track_sync_device(torch.from_numpy(t2).to(dev), 80, torch.from_numpy(ok).to(dev), torch.from_numpy(valid).to(dev),
                                           torch.from_numpy(nfr).to(dev), Cn, F, d_cell, d_scr, d_rx, d_t)
        torch.cuda.synchronize()
        assert np.array_equal(d_scr.cpu().numpy().view(np.uint32), w_scr), call
        assert np.array_equal(d_rx.cpu().numpy().view(np.uint32), w_rx), call
        assert np.array_equal(d_t.cpu().numpy().view(np.uint32), w_t), call
        got = d_cell.cpu().numpy().view(np.uint32)
        for c in range(Cn):
            w = want[c]
            assert list(got[c]) == [w["scr"], w["cc"], w["mcc"], w["mnc"], *w["tcd"], w["phy"].tn, w["phy"].fn, w["phy"].mn], (call, c)
    assert max(w["phy"].mn for w in want) > 0 and any(w["scr"] for w in want)


@pytest.mark.gpu
def test_gpu_decode_frames_all_kinds_in_one_launch(pkg, lref, oracle):
    """tetra_burst_index_device + tetra_lmac_decode_frames_device on the GPU: the four frame lists equal numpy's, and ONE launch with
    a job per kind (SCH/F, SB2, NDB 1 + 2, BBK, SB1; counted rows, labels) gives, row for row, what the restated tetra_burst_rx_cb
    split followed by the REFERENCE's decoding chain gives for the listed frames -- incl. workgroups that end inside a list."""
    import torch
    lb, bb = pkg.lmac_binding, pkg.bsync_binding
    dev = torch.device("cuda", 0)
    F = 25
    n = 40 * F
    frames, packed, types, codes = make_frames(lref, oracle, n, 33)
    d_fr = torch.from_numpy(packed.astype(np.int64)).to(dev).to(torch.int32).contiguous()        # uint32 bit patterns
    d_ft = torch.from_numpy(types).to(dev)
    d_codes = torch.from_numpy(codes.astype(np.int64)).to(dev).to(torch.int32)
    lists = torch.full((4, n), -7, dtype=torch.int32, device=dev)
    counts = torch.zeros(4, dtype=torch.int32, device=dev)
    chan_first = torch.zeros((4, n // F), dtype=torch.int32, device=dev)
    bb.index_device(d_ft, F, lists, counts, chan_first)
    torch.cuda.synchronize()
    want_lists = [np.flatnonzero(types == 3), np.flatnonzero(types == 0), np.flatnonzero(types == 1), np.flatnonzero(np.isin(types, (0, 1, 3)))]
    for k in range(4):
        assert int(counts[k]) == want_lists[k].size and np.array_equal(lists[k, :want_lists[k].size].cpu().numpy(), want_lists[k])
        assert np.array_equal(chan_first[k].cpu().numpy(), np.searchsorted(want_lists[k], np.arange(0, n, F)))
    bitnum = torch.arange(n, dtype=torch.int32, device=dev) * 510 + 7
    t_rx = torch.arange(n, dtype=torch.int32, device=dev) + 1000
    t_af = torch.arange(n, dtype=torch.int32, device=dev) + 5000
    kinds = ((5, 0, 1), (1, 2, 0), (2, 1, 2), (2, 2, 2), (3, 0, 3), (0, 1, 0))       # (tpsap, blk, list)
    jobs, outs = [], []
    for tpsap, blk, li in kinds:
        n2 = 32 if tpsap == 3 else lref.BLK_PARAM[tpsap][1]
        t2 = torch.full((n, n2 + 8), 5, dtype=torch.uint8, device=dev)
        ok = torch.full((n,), -3, dtype=torch.int32, device=dev)
        lab = torch.full((n, 6), -1, dtype=torch.int32, device=dev)
        outs.append((t2, ok, lab))
        jobs.append(dict(type=tpsap, blk_num=blk, row_frame=lists[li], n_rows=counts[li:li + 1], max_rows=n, out_stride=n2 + 8,
                         frame_scramb=None if tpsap == 0 else d_codes, type2=t2, crc_ok=ok, labels=lab))
    lb.decode_frames_device(d_fr, d_ft, jobs, F, bitnum, t_rx, t_af)
    torch.cuda.synchronize()
    # the same launch with the caller's decision scratch instead of the library's pool, and an SB1-only launch
    need = lb.decode_frames_workspace_bytes(jobs[:5])
    assert need == 2 * sum((lref.BLK_PARAM[t][1] + 4) * 64 * ((n + 63) // 64) for t, _, _ in kinds[:5] if t != 3) and lb.decode_frames_workspace_bytes(jobs[4:5]) == 0
    again = [tuple(torch.full_like(x, 9) for x in o) for o in outs]
    jobs2 = [dict(j, type2=o[0], crc_ok=o[1], labels=o[2]) for j, o in zip(jobs, again)]
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    lb.decode_frames_device(d_fr, d_ft, jobs2[:5], F, bitnum, t_rx, t_af, d_workspace=ws)
    lb.decode_frames_device(d_fr, d_ft, jobs2[5:], F, bitnum, t_rx, t_af)
    with pytest.raises(pkg.TetraDemodError):
        lb.decode_frames_device(d_fr, d_ft, jobs2[:5], F, bitnum, t_rx, t_af, d_workspace=ws[: need - 256])       # TETRA_ERR_SIZE
    torch.cuda.synchronize()
    for (tpsap, blk, li), o, o2 in zip(kinds, outs, again):
        cnt = want_lists[li].size
        n2 = 32 if tpsap == 3 else lref.BLK_PARAM[tpsap][1]
        assert torch.equal(o[0][:cnt, :n2], o2[0][:cnt, :n2]) and torch.equal(o[1][:cnt], o2[1][:cnt]) and torch.equal(o[2][:cnt], o2[2][:cnt]), (tpsap, blk)
    good = 0
    for (tpsap, blk, li), (t2, ok, lab) in zip(kinds, outs):
        rows = want_lists[li]
        n2 = 32 if tpsap == 3 else lref.BLK_PARAM[tpsap][1]
        n345 = 30 if tpsap == 3 else lref.BLK_PARAM[tpsap][0]
        t2, ok, lab = t2.cpu().numpy(), ok.cpu().numpy(), lab.cpu().numpy()
        assert (t2[rows.size:] == 5).all() and (ok[rows.size:] == -3).all() and (lab[rows.size:] == -1).all()      # rows past the count: untouched
        assert (t2[:rows.size, n2:] == 5).all()                                                                   # and the rows' padding
        for j, r in enumerate(rows):
            row = oracle.bsync_demux(frames[r], int(types[r]), tpsap, blk)
            t5 = np.zeros(n345, np.uint8)
            t5[:row.size] = row
            want, want_ok = lref.lmac_decode(tpsap, t5, lref.SCRAMB_INIT if tpsap == 0 else int(codes[r]))
            if tpsap == 3:
                want = np.concatenate([want, np.zeros(2, np.uint8)])
            assert np.array_equal(t2[j, :n2], want) and ok[j] == want_ok, (tpsap, blk, j)
            assert list(lab[j]) == [r // F, r % F, r * 510 + 7, r + 1000, r + 5000, want_ok]
            good += int(want_ok) if tpsap != 3 else 0
    assert good > 150
    # counts given on the host instead (d_n_rows = NULL: max_rows rows exist), lists without the per-channel positions, an empty call
    li = 1
    cnt = want_lists[li].size
    t2b = torch.full((cnt, 296), 9, dtype=torch.uint8, device=dev)
    okb = torch.full((cnt,), -3, dtype=torch.int32, device=dev)
    lb.decode_frames_device(d_fr, d_ft, [dict(type=5, blk_num=0, row_frame=lists[li], n_rows=None, max_rows=cnt, out_stride=296, frame_scramb=d_codes,
                                              type2=t2b, crc_ok=okb)])
    torch.cuda.synchronize()
    assert torch.equal(t2b[:, :288], outs[0][0][:cnt, :288]) and torch.equal(okb, outs[0][1][:cnt])
    lists2 = torch.full((4, n), -7, dtype=torch.int32, device=dev)
    counts2 = torch.full((4,), -1, dtype=torch.int32, device=dev)
    bb.index_device(d_ft, F, lists2, counts2, None)
    bb.index_device(d_ft[:0], F, torch.empty((4, 1), dtype=torch.int32, device=dev), counts, None)      # n = 0: counts zeroed, nothing launched
    torch.cuda.synchronize()
    assert torch.equal(counts2.cpu(), torch.tensor([w.size for w in want_lists], dtype=torch.int32)) and int(counts.abs().sum()) == 0
    for k in range(4):
        assert torch.equal(lists2[k, :want_lists[k].size], lists[k, :want_lists[k].size])
    # argument errors: statuses, not launches
    from ctypes import byref
    L = pkg.binding.load_library()
    src = lb.Frames(d_fr.data_ptr(), d_ft.data_ptr(), n, F, None, None, None, None, 0)
    t2, ok, lab = outs[0]
    def one(**kw):
        base = dict(type=5, blk_num=0, d_row_frame=lists[1].data_ptr(), d_n_rows=None, max_rows=n, out_stride=296, d_frame_scramb=d_codes.data_ptr(),
                    d_type2=t2.data_ptr(), d_crc_ok=ok.data_ptr(), d_labels=None)
        base.update(kw)
        return L.tetra_lmac_decode_frames_device(byref(src), byref(lb.Job(*[base[f] for f, _ in lb.Job._fields_])), 1, None)
    ERR_ARG, ERR_SIZE, ERR_ALIGN = -1, -6, -7                           # include/tetra_demod.h
    assert one(type=4) == ERR_ARG                    # SCH/HU: no downlink burst carries it
    assert one(type=0, blk_num=2) == ERR_ARG         # SB1 is block 1
    assert one(out_stride=280) == ERR_SIZE
    assert one(out_stride=292) == ERR_ALIGN
    assert one(d_frame_scramb=None) == ERR_ARG
    assert one(d_labels=lab.data_ptr()) == ERR_ARG   # labels without the per-frame arrays
    assert one(max_rows=0) == 0
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_gpu_track_sync_lists_equals_the_slot_layout_tracker(pkg, lref):
    """tetra_lmac_track_sync_lists_device (compact SB1 rows, a wavefront per channel) == tetra_lmac_track_sync_device (slot layout,
    itself checked against the reference's field read-out and TDMA arithmetic in tests/test_burst_sync.py): cell state, per-slot
    codes and times, over two calls with carried state, channels without any SYNC burst, bad CRCs, short frame counts; plus the
    SB1 rows' labels."""
    import torch
    lb, bb = pkg.lmac_binding, pkg.bsync_binding
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(9)
    C_, F = 37, 150                                      # three 64-slot groups per channel, the last one ragged
    n = C_ * F
    cell0 = np.zeros((C_, 10), np.int32)
    cell0[7:20] = rng.integers(0, 70, (13, 10))          # arbitrary carried states (clock digits out of range): the wrap tests normalise them
    cell0[20, 4:] = [100, 200, 500, 7, 36, 121]
    cell_a = torch.from_numpy(cell0).to(dev)
    cell_b = torch.from_numpy(cell0.copy()).to(dev)
    for call in range(2):
        types = rng.choice(np.array([0, 1, 3, 3, -1, -2], np.int32), n)
        types.reshape(C_, F)[5] = 0                                     # a channel without SYNC bursts
        nf = rng.integers(F - 6, F + 1, C_).astype(np.int32)
        nf[3], nf[4], nf[6] = 0, 64, 70
        sync = np.flatnonzero(types == 3)
        t2c = rng.integers(0, 2, (sync.size, 80), dtype=np.uint8)
        okc = (rng.random(sync.size) < 0.8).astype(np.int32)
        # slot layout for the old tracker
        slot_t2 = np.zeros((n, 80), np.uint8)
        slot_ok = np.zeros(n, np.int32)
        slot_valid = np.zeros(n, np.int32)
        slot_t2[sync], slot_ok[sync], slot_valid[sync] = t2c, okc, 1
        d = lambda a: torch.from_numpy(a).to(dev)
        outs_a = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3)]
        outs_b = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3)]
        lb.track_sync_device(d(slot_t2), 80, d(slot_ok), d(slot_valid), d(nf), C_, F, cell_a, *outs_a)
        d_ft = d(types)
        lists = torch.zeros((4, n), dtype=torch.int32, device=dev)
        counts = torch.zeros(4, dtype=torch.int32, device=dev)
        chan_first = torch.zeros((4, C_), dtype=torch.int32, device=dev)
        bb.index_device(d_ft, F, lists, counts, chan_first)
        bitnum = torch.arange(n, dtype=torch.int32, device=dev) * 3
        labels = torch.full((n, 6), -1, dtype=torch.int32, device=dev)
        lb.track_sync_lists_device(d(t2c), 80, d(okc), d_ft, d(nf), chan_first[0], C_, F, cell_b, *outs_b, d_frame_bitnum=bitnum, d_sb1_labels=labels)
        torch.cuda.synchronize()
        assert torch.equal(cell_a, cell_b)
        for a, b in zip(outs_a, outs_b):
            assert torch.equal(a, b)
        lab = labels.cpu().numpy()
        trx, taf = outs_b[1].cpu().numpy(), outs_b[2].cpu().numpy()
        for j, r in enumerate(sync):
            c, f = divmod(int(r), F)
            if f < nf[c]:
                assert list(lab[j]) == [c, f, 3 * r, trx[r], taf[r], okc[j]]
            else:
                assert (lab[j] == -1).all()              # a SYNC-typed slot past the channel's frame count is not a frame
        assert (lab[sync.size:] == -1).all()
    # d_n_frames = NULL: every frame slot of every channel counts
    types = rng.choice(np.array([0, 1, 3, -1], np.int32), n)
    sync = np.flatnonzero(types == 3)
    t2c = rng.integers(0, 2, (sync.size, 80), dtype=np.uint8)
    okc = np.ones(sync.size, np.int32)
    slot_t2 = np.zeros((n, 80), np.uint8)
    slot_ok, slot_valid = np.zeros(n, np.int32), np.zeros(n, np.int32)
    slot_t2[sync], slot_ok[sync], slot_valid[sync] = t2c, okc, 1
    d = lambda a: torch.from_numpy(a).to(dev)
    d_ft = d(types)
    lists = torch.zeros((4, n), dtype=torch.int32, device=dev)
    counts = torch.zeros(4, dtype=torch.int32, device=dev)
    chan_first = torch.zeros((4, C_), dtype=torch.int32, device=dev)
    bb.index_device(d_ft, F, lists, counts, chan_first)
    oa = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3)]
    ob = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3)]
    lb.track_sync_device(d(slot_t2), 80, d(slot_ok), d(slot_valid), None, C_, F, cell_a, *oa)
    lb.track_sync_lists_device(d(t2c), 80, d(okc), d_ft, None, chan_first[0], C_, F, cell_b, *ob)
    torch.cuda.synchronize()
    assert torch.equal(cell_a, cell_b) and all(torch.equal(a, b) for a, b in zip(oa, ob))
