"""Binding of oracle/_ref/libtetra_burst_ref.so -- the REFERENCE's own phy/tetra_burst.c + phy/tetra_burst_sync.c,
compiled from /root/reference where they lie by oracle/build_ref.sh (test infrastructure; never shipped, never copied).

Gives the tests the reference's real tetra_find_train_seq() (tetra_burst.c:271-341) and its burst builders
build_sync_c_d_burst() / build_norm_c_d_burst() (tetra_burst.c:171-269).  The library keeps undefined references into
the lower MAC (tetra_burst_rx_cb -> tp_sap_udata_ind); they are never called, so it is loaded with lazy binding."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libtetra_burst_ref.so")

TRAIN_NORM_1, TRAIN_NORM_2, TRAIN_NORM_3, TRAIN_SYNC, TRAIN_EXT = 0, 1, 2, 3, 4
ALL_MASK = 0x1f

_lib = None


def build():
    """Rebuild from /root/reference when it is present (no-op on the GPU box, which carries the prebuilt file)."""
    subprocess.run(["sh", os.path.join(_HERE, "build_ref.sh")], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def available():
    if not os.path.exists(LIB_PATH):
        try:
            build()
        except Exception:
            return False
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref is not built and /root/reference is not present")
        L = C.CDLL(LIB_PATH, mode=os.RTLD_LAZY)
        vp = C.c_void_p
        L.tetra_find_train_seq.argtypes = [vp, C.c_uint, C.c_uint32, C.POINTER(C.c_uint)]
        L.tetra_find_train_seq.restype = C.c_int
        L.build_norm_c_d_burst.argtypes = [vp, vp, vp, vp, C.c_int]
        L.build_norm_c_d_burst.restype = C.c_int
        L.build_sync_c_d_burst.argtypes = [vp, vp, vp, vp]
        L.build_sync_c_d_burst.restype = C.c_int
        _lib = L
    return _lib


def find_train_seq(bits, end_of_in, mask=ALL_MASK):
    """The reference's tetra_find_train_seq on one row -> (type or -1, offset or -1).  `bits` must extend at least
    21 bytes past end_of_in (the reference reads its look-ahead there)."""
    b = np.ascontiguousarray(bits, np.uint8)
    assert b.size >= end_of_in + 21
    off = C.c_uint(0xffffffff)
    t = lib().tetra_find_train_seq(b.ctypes.data_as(C.c_void_p), int(end_of_in), int(mask), C.byref(off))
    return (t, int(off.value)) if t >= 0 else (-1, -1)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build_norm_burst(bkn1, bb, bkn2, two_log_chan=0):
    """510-bit normal continuous downlink burst (training sequence 1 or 2 at bit 244)."""
    buf = np.zeros(600, np.uint8)
    n = lib().build_norm_c_d_burst(_p(buf), _p(np.ascontiguousarray(bkn1, np.uint8)), _p(np.ascontiguousarray(bb, np.uint8)),
                                   _p(np.ascontiguousarray(bkn2, np.uint8)), int(two_log_chan))
    return buf[:n].copy()


def build_sync_burst(sb, bb, bkn):
    """510-bit synchronisation continuous downlink burst (sync training sequence at bit 214)."""
    buf = np.zeros(600, np.uint8)
    n = lib().build_sync_c_d_burst(_p(buf), _p(np.ascontiguousarray(sb, np.uint8)), _p(np.ascontiguousarray(bb, np.uint8)),
                                   _p(np.ascontiguousarray(bkn, np.uint8)))
    return buf[:n].copy()


# ---------------------------------------------------------------------------------------------------------------------
# The reference's synchroniser RUN: tetra_burst_sync_in() (phy/tetra_burst_sync.c:54-155) -> tetra_burst_rx_cb()
# (phy/tetra_burst.c:343-393) -> tp_sap_udata_ind(), the last being the test-side recorder tests/refrec/tp_sap_recorder.c
# (oracle/_ref/libtetra_tpsap_recorder.so).  The recorder library is loaded RTLD_GLOBAL so that the reference library's lazy
# reference to tp_sap_udata_ind binds to it.
# ---------------------------------------------------------------------------------------------------------------------
RECORDER_LIB_PATH = os.path.join(_HERE, "_ref", "libtetra_tpsap_recorder.so")
_rec = None


class RecEvent(C.Structure):
    _fields_ = [("type", C.c_int32), ("blk_num", C.c_int32), ("len", C.c_int32), ("frame_bitnum", C.c_uint32),
                ("tn", C.c_uint32), ("fn", C.c_uint32), ("mn", C.c_uint32), ("bits", C.c_uint8 * 432)]


def sync_run_available():
    if not available():
        return False
    if not os.path.exists(RECORDER_LIB_PATH):
        try:
            build()
        except Exception:
            return False
    if not os.path.exists(RECORDER_LIB_PATH):
        return False
    try:
        lib().tetra_tdma_time_add_tn      # libraries built before tetra_tdma.c was added cannot run the state machine
    except AttributeError:
        return False
    return True


def _rec_lib():
    global _rec
    if _rec is None:
        if not sync_run_available():
            raise RuntimeError("oracle/_ref recorder not built and /root/reference not present")
        R = C.CDLL(RECORDER_LIB_PATH, mode=os.RTLD_GLOBAL | os.RTLD_LAZY)
        vp = C.c_void_p
        R.rec_new.argtypes = [vp]
        R.rec_new.restype = vp
        R.rec_free.argtypes = [vp]
        R.rec_set_traffic.argtypes = [vp, C.c_int]
        R.rec_feed.argtypes = [vp, vp, vp, C.c_int, C.c_int]
        R.rec_rx_state.argtypes = [vp, vp, vp]
        R.rec_event_count.argtypes = [vp]
        R.rec_event_count.restype = C.c_int
        R.rec_event_size.restype = C.c_int
        R.rec_events.argtypes = [vp, C.c_int, C.c_int, vp]
        R.rec_clear_events.argtypes = [vp]
        assert R.rec_event_size() == C.sizeof(RecEvent)
        _rec = R
    return _rec


class ReferenceBurstSync:
    """One tetra_rx_state driven through the REFERENCE's tetra_burst_sync_in.  feed() returns the tp_sap_udata_ind calls
    the reference made during it: list of (type, blk_num, bits uint8[len], frame_bitnum).  NOTE the reference keeps the
    slot counter in a global (t_phy_state): instances share it, exactly like the plugin's single decoder."""

    def __init__(self, is_traffic=0):
        R, L = _rec_lib(), lib()
        phy = C.addressof(C.c_char.in_dll(L, "t_phy_state"))
        self._h = R.rec_new(phy)
        R.rec_set_traffic(self._h, int(is_traffic))
        self._sync_in = C.cast(L.tetra_burst_sync_in, C.c_void_p)

    def close(self):
        if self._h:
            _rec_lib().rec_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def feed(self, bits, chunk=1):
        R = _rec_lib()
        b = np.ascontiguousarray(bits, np.uint8)
        R.rec_clear_events(self._h)
        R.rec_feed(self._h, self._sync_in, b.ctypes.data_as(C.c_void_p), int(b.size), int(chunk))
        n = R.rec_event_count(self._h)
        ev = (RecEvent * max(n, 1))()
        if n:
            R.rec_events(self._h, 0, n, ev)
        return [(int(e.type), int(e.blk_num), np.frombuffer(bytes(e.bits), np.uint8)[: e.len].copy(), int(e.frame_bitnum),
                 (int(e.tn), int(e.fn), int(e.mn))) for e in ev[:n]]

    @property
    def state(self):
        """(state, bits_in_buf, bitbuf_start_bitnum, next_frame_start_bitnum) like oracle.BurstSyncOracle.state"""
        w = (C.c_uint32 * 4)()
        _rec_lib().rec_rx_state(self._h, w, None)
        return int(w[0]), int(w[1]), int(w[2]), int(w[3])

    def bitbuf(self):
        w = (C.c_uint32 * 4)()
        buf = np.zeros(4096, np.uint8)
        _rec_lib().rec_rx_state(self._h, w, buf.ctypes.data_as(C.c_void_p))
        return buf[: int(w[1])].copy()


# what tetra_burst_rx_cb hands downstream per burst type, in its call order (tetra_burst.c:353-384): (tp_sap type, blk_num)
RX_CB_BLOCKS = {
    TRAIN_SYNC: ((0, 1), (3, 0), (1, 2)),       # SB1/BLK_1, BBK/0, SB2/BLK_2
    TRAIN_NORM_2: ((3, 0), (2, 1), (2, 2)),     # BBK/0, NDB/BLK_1, NDB/BLK_2
    TRAIN_NORM_1: ((3, 0), (5, 0)),             # BBK/0, SCH_F/0
}


# ---------------------------------------------------------------------------------------------------------------------
# Lower-MAC channel coding: oracle/_ref/libtetra_lmac_ref.so = the reference's own
# lower_mac/{tetra_scramb,tetra_interleave,tetra_conv_enc,crc_simple,viterbi,viterbi_cch,osmo_conv}.c
# ---------------------------------------------------------------------------------------------------------------------
LMAC_LIB_PATH = os.path.join(_HERE, "_ref", "libtetra_lmac_ref.so")
# enum tp_sap_data_type, phy/tetra_burst.h:9-16
TPSAP_T_SB1, TPSAP_T_SB2, TPSAP_T_NDB, TPSAP_T_BBK, TPSAP_T_SCH_HU, TPSAP_T_SCH_F = range(6)
# tetra_blk_param[], lower_mac/tetra_lower_mac.c:58-105: (type345_bits, type2_bits, type1_bits, interleave_a, have_crc16)
BLK_PARAM = {
    TPSAP_T_SB1: (120, 80, 60, 11, 1),
    TPSAP_T_SB2: (216, 144, 124, 101, 1),
    TPSAP_T_NDB: (216, 144, 124, 101, 1),
    TPSAP_T_BBK: (30, 30, 14, 0, 0),
    TPSAP_T_SCH_HU: (168, 112, 92, 13, 1),
    TPSAP_T_SCH_F: (432, 288, 268, 103, 1),
}
SCRAMB_INIT = 3            # lower_mac/tetra_scramb.h:14
TETRA_CRC_OK = 0x1d0f      # tetra_common.h:330
RCPC_PUNCT_2_3 = 0         # enum tetra_rcpc_puncturer, first entry (lower_mac/tetra_conv_enc.h)

_lmac = None


def lmac_available():
    if not os.path.exists(LMAC_LIB_PATH):
        try:
            build()
        except Exception:
            return False
    return os.path.exists(LMAC_LIB_PATH)


def lmac_lib():
    global _lmac
    if _lmac is None:
        if not lmac_available():
            raise RuntimeError("oracle/_ref/libtetra_lmac_ref.so is not built and /root/reference is not present")
        L = C.CDLL(LMAC_LIB_PATH)
        vp = C.c_void_p
        L.tetra_scramb_bits.argtypes = [C.c_uint32, vp, C.c_int]
        L.tetra_scramb_get_init.argtypes = [C.c_uint16, C.c_uint16, C.c_uint8]
        L.tetra_scramb_get_init.restype = C.c_uint32
        L.block_interleave.argtypes = [C.c_uint32, C.c_uint32, vp, vp]
        L.block_interleave.restype = None
        L.block_deinterleave.argtypes = [C.c_uint32, C.c_uint32, vp, vp]
        L.block_deinterleave.restype = None
        L.tetra_rcpc_depunct.argtypes = [C.c_int, vp, C.c_uint32, vp]
        L.get_punctured_rate.argtypes = [C.c_int, vp, C.c_uint32, vp]
        L.viterbi_dec_sb1_wrapper.argtypes = [vp, vp, C.c_uint]
        L.viterbi_dec_sb1_wrapper.restype = None
        L.crc16_ccitt_bits.argtypes = [vp, C.c_uint]
        L.crc16_ccitt_bits.restype = C.c_uint16
        L.conv_enc_init.argtypes = [vp]
        L.conv_enc_input.argtypes = [vp, vp, C.c_int, vp]
        L.tetra_punct_test.restype = C.c_int
        _lmac = L
    return _lmac


def scramb_get_init(mcc, mnc, colour):
    return int(lmac_lib().tetra_scramb_get_init(int(mcc), int(mnc), int(colour)))


def lmac_decode(blk_type, type5, scramb_init):
    """One block through the reference's primitives in the order tp_sap_udata_ind calls them
    (lower_mac/tetra_lower_mac.c:181-227) -> (type2 bits [type2_bits], crc_ok)."""
    L = lmac_lib()
    n345, n2, n1, a, have_crc = BLK_PARAM[blk_type]
    type4 = np.zeros(512, np.uint8)
    type4[:n345] = np.asarray(type5, np.uint8)[:n345]                                   # :184 memcpy
    L.tetra_scramb_bits(SCRAMB_INIT if blk_type == TPSAP_T_SB1 else int(scramb_init) & 0xffffffff, _p(type4), n345)
    type2 = np.zeros(512, np.uint8)
    if a:
        type3 = np.zeros(512, np.uint8)
        L.block_deinterleave(n345, a, _p(type4), _p(type3))                             # :204
        type3dp = np.full(512 * 4, 0xff, np.uint8)                                      # :208 memset 0xff
        L.tetra_rcpc_depunct(RCPC_PUNCT_2_3, _p(type3), n345, _p(type3dp))              # :209
        L.viterbi_dec_sb1_wrapper(_p(type3dp), _p(type2), n2)                           # :212
    crc_ok = 0
    if have_crc:
        crc_ok = int(L.crc16_ccitt_bits(_p(type2), n1 + 16) == TETRA_CRC_OK)            # :218-220
    elif blk_type == TPSAP_T_BBK:
        crc_ok = 1                                                                      # :231-234 (RM decode is a FIXME)
        type2[:n2] = type4[:n2]
    return type2[:n2].copy(), crc_ok


def lmac_encode(blk_type, type1, scramb_init):
    """Transmit side built from the reference's own encoder primitives (conv_enc_input, get_punctured_rate,
    block_interleave, tetra_scramb_bits; EN 300 392-2 8.2): type1 bits -> type5 bits.  Used to make known-answer
    blocks; the CRC16 appended is the ones' complement of crc16_ccitt_bits over the type1 bits, MSB first."""
    L = lmac_lib()
    n345, n2, n1, a, have_crc = BLK_PARAM[blk_type]
    assert have_crc and a
    t1 = np.ascontiguousarray(type1, np.uint8)[:n1]
    crc = (~int(L.crc16_ccitt_bits(_p(t1.copy()), n1))) & 0xffff
    type2 = np.zeros(n2, np.uint8)
    type2[:n1] = t1
    type2[n1:n1 + 16] = [(crc >> (15 - i)) & 1 for i in range(16)]                       # + 4 zero tail bits
    ces = np.zeros(64, np.uint8)
    L.conv_enc_init(_p(ces))
    mother = np.zeros(4 * n2, np.uint8)
    L.conv_enc_input(_p(ces), _p(type2), n2, _p(mother))
    type3 = np.zeros(n345, np.uint8)
    L.get_punctured_rate(RCPC_PUNCT_2_3, _p(mother), n345, _p(type3))
    type4 = np.zeros(n345, np.uint8)
    L.block_interleave(n345, a, _p(type3), _p(type4))
    L.tetra_scramb_bits(SCRAMB_INIT if blk_type == TPSAP_T_SB1 else int(scramb_init) & 0xffffffff, _p(type4), n345)
    return type4
