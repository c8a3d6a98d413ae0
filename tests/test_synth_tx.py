"""The coded-downlink generator of synth.py (transmit side of the lower MAC + the two continuous downlink bursts) against the
reference's own encoder primitives and burst builders compiled from /root/reference (oracle/_ref): bit for bit."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def lref(ref):
    if not ref.lmac_available():
        pytest.skip("oracle/_ref/libtetra_lmac_ref.so not built and /root/reference not present")
    return ref


def test_scrambling_code_and_sequence(synth, lref):
    for cell in ((262, 1, 5), (1023, 16383, 63), (0, 0, 0), (234, 14, 1)):
        assert synth.tx_scramb_code(*cell) == lref.scramb_get_init(*cell)
    codes = np.array([3, 0x41800117, 0xffffffff, 12345], np.uint32)
    seq = synth.tx_scramb_seq(codes, 432)
    for i, c in enumerate(codes):
        z = np.zeros(432, np.uint8)
        lref.lmac_lib().tetra_scramb_bits(int(c), z.ctypes.data, 432)
        assert np.array_equal(seq[i], z)


@pytest.mark.parametrize("kind,blk", [("sb1", 0), ("sb2", 1), ("ndb", 2), ("schf", 5)])
def test_encoder_equals_reference_primitives(synth, lref, kind, blk):
    rng = np.random.default_rng(blk)
    n1 = synth.TX_BLK[kind][2]
    t1 = rng.integers(0, 2, (40, n1), dtype=np.uint8)
    codes = rng.integers(0, 2 ** 32, 40, dtype=np.uint64).astype(np.uint32)
    t5 = synth.tx_encode(kind, t1, codes)
    for r in range(40):
        want = lref.lmac_encode(blk, t1[r], int(codes[r]))
        assert np.array_equal(t5[r], want), r
        t2, ok = lref.lmac_decode(blk, t5[r], int(codes[r]))          # and the reference decodes it back, CRC good
        assert ok == 1 and np.array_equal(t2[:n1], t1[r])


def test_bursts_equal_reference_builders(synth, lref):
    """Everything but the four phase-adjustment bits (12..13, 498..499): the reference indexes its phase table without the offset its
    own PHASE() macro provides (phy/tetra_burst.c:162 against :110-121), so its h-bits are not the standard's; they carry no
    information and no receiver stage reads them."""
    rng = np.random.default_rng(4)
    keep = np.ones(510, bool)
    keep[[12, 13, 498, 499]] = False
    for _ in range(20):
        sb, bb, bkn = rng.integers(0, 2, 120), rng.integers(0, 2, 30), rng.integers(0, 2, 216)
        assert np.array_equal(synth.tx_sync_burst(sb, bb, bkn)[keep], lref.build_sync_burst(sb, bb, bkn)[keep])
        b1 = rng.integers(0, 2, 216)
        for two in (0, 1):
            assert np.array_equal(synth.tx_norm_burst(b1, bb, bkn, two)[keep], lref.build_norm_burst(b1, bb, bkn, two)[keep])


def test_phase_adjustment_bits_close_the_phase(synth):
    """9.4.4.3.6: with the h-bits in place the phase accumulated over each adjusted range is a multiple of 2 pi."""
    rng = np.random.default_rng(5)
    step = {(0, 0): 1, (0, 1): 3, (1, 1): -3, (1, 0): -1}
    def total(b, n1, n2):
        return sum(step[(int(b[2 * k]), int(b[2 * k + 1]))] for k in range(n1 - 1, n2))
    for _ in range(10):
        s = synth.tx_sync_burst(rng.integers(0, 2, 120), rng.integers(0, 2, 30), rng.integers(0, 2, 216))
        # hc (symbol 7) closes 8..108, hd (symbol 250) closes 109..249
        assert (total(s, 7, 108)) % 8 == 0 and (total(s, 109, 250)) % 8 == 0
        n = synth.tx_norm_burst(rng.integers(0, 2, 216), rng.integers(0, 2, 30), rng.integers(0, 2, 216), 0)
        assert (total(n, 7, 122)) % 8 == 0 and (total(n, 123, 250)) % 8 == 0


def test_downlink_is_received_by_the_reference(synth, ref, lref):
    """gen_downlink's bit stream through the REFERENCE's own tetra_burst_sync_in -> tetra_burst_rx_cb (oracle/_ref, one bit per
    call) and the reference's decoder primitives: every block the reference hands downstream decodes with a good CRC to the type-1
    bits the generator says it sent, the SYNC PDU fields included."""
    if not ref.sync_run_available():
        pytest.skip("oracle/_ref recorder not available")
    cell = (901, 77, 33)
    bits, sent = synth.gen_downlink(24, 7, cell=cell)
    code = lref.scramb_get_init(*cell)
    rx = ref.ReferenceBurstSync()
    ev = rx.feed(np.concatenate([bits, np.zeros(600, np.uint8)]), chunk=1)
    rx.close()
    want = {k: {tuple(v.tolist()) for _, v in sent[k]} for k in sent}
    seen = {k: 0 for k in want}
    for typ, blk, b, bitnum, _ in ev:
        if typ == lref.TPSAP_T_BBK:
            t2, _ = lref.lmac_decode(typ, b, code)
            assert tuple(t2[:30].tolist()) in want["bbk"]
            seen["bbk"] += 1
            continue
        t2, ok = lref.lmac_decode(typ, b, code)
        name = {(0, 1): "sb1", (1, 2): "sb2", (2, 1): "ndb1", (2, 2): "ndb2", (5, 0): "schf"}[(typ, blk)]
        n1 = lref.BLK_PARAM[typ][2]
        assert ok == 1 and tuple(t2[:n1].tolist()) in want[name], (typ, blk, bitnum)
        seen[name] += 1
    assert seen["sb1"] >= 5 and seen["sb2"] >= 5 and seen["schf"] >= 10 and seen["ndb1"] >= 5 and seen["ndb2"] >= 5 and seen["bbk"] >= 20
