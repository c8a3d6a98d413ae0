"""ctypes binding of the receive chain behind one handle (include/tetra_rx.h)."""
import ctypes as C

import numpy as np

from . import binding as B
from .binding import TetraDemodError, load_library
from .bsync_binding import BsyncState

RX_EXPORTS = ["tetra_rx_default_config", "tetra_rx_create", "tetra_rx_destroy", "tetra_rx_reset", "tetra_rx_process_device",
              "tetra_rx_process", "tetra_rx_wait", "tetra_rx_max_rows", "tetra_rx_type1_bits", "tetra_rx_fetch", "tetra_rx_rows_device",
              "tetra_rx_get_cell", "tetra_rx_get_sync_state", "tetra_rx_bits_device", "tetra_rx_demod", "tetra_rx_stage_ms"]
KIND_SB1, KIND_BBK, KIND_SB2, KIND_NDB1, KIND_NDB2, KIND_SCH_F = range(6)
N_KINDS = 6
FLAG_ONE_STREAM = 1


class RxConfig(C.Structure):
    _fields_ = [("demod", B.Config), ("kinds", C.c_int32), ("flags", C.c_int32)]


class RxBlock(C.Structure):
    _fields_ = [("channel", C.c_int32), ("frame_slot", C.c_int32), ("bitnum", C.c_uint32), ("tdma_time_rx", C.c_uint32),
                ("tdma_time", C.c_uint32), ("crc_ok", C.c_int32)]


BLOCK_DTYPE = np.dtype([("channel", "<i4"), ("frame_slot", "<i4"), ("bitnum", "<u4"), ("tdma_time_rx", "<u4"), ("tdma_time", "<u4"),
                        ("crc_ok", "<i4")])


class CellState(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("scramb_init", "colour_code", "mcc", "mnc", "tcd_tn", "tcd_fn", "tcd_mn", "phy_tn", "phy_fn",
                                          "phy_mn")]


_ready = False


def _lib():
    global _ready
    L = load_library()
    if not _ready:
        vp, i32 = C.c_void_p, C.c_int
        L.tetra_rx_default_config.argtypes = [C.POINTER(RxConfig)]
        L.tetra_rx_create.argtypes = [C.POINTER(RxConfig), C.POINTER(vp)]
        L.tetra_rx_destroy.argtypes = [vp]
        L.tetra_rx_reset.argtypes = [vp]
        L.tetra_rx_process_device.argtypes = [vp, vp, i32, vp]
        L.tetra_rx_process.argtypes = [vp, vp, i32]
        L.tetra_rx_wait.argtypes = [vp]
        L.tetra_rx_max_rows.argtypes = [vp]
        L.tetra_rx_type1_bits.argtypes = [i32]
        L.tetra_rx_fetch.argtypes = [vp, i32, i32, vp, vp, i32, i32, C.POINTER(i32)]
        L.tetra_rx_rows_device.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(vp), C.POINTER(vp), vp]
        L.tetra_rx_get_cell.argtypes = [vp, i32, i32, vp]
        L.tetra_rx_get_sync_state.argtypes = [vp, i32, i32, vp]
        L.tetra_rx_bits_device.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(vp), vp]
        L.tetra_rx_demod.argtypes = [vp]
        L.tetra_rx_demod.restype = vp
        L.tetra_rx_stage_ms.argtypes = [vp, C.POINTER(C.c_float * 4)]
        for n in RX_EXPORTS:
            if n != "tetra_rx_demod":
                getattr(L, n).restype = i32
        _ready = True
    return L


def type1_bits(kind):
    return int(_lib().tetra_rx_type1_bits(int(kind)))


def _stream_ptr(stream):
    if stream is None:
        return None
    return C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))


class RxChain:
    """IQ in, decoded type-1 blocks + CRC + TDMA time + cell state out, for C channels on one GPU: demodulator -> burst
    synchroniser -> demultiplexer -> lower-MAC decoder -> SYNC-PDU tracker, ordered and buffered by the library."""

    def __init__(self, n_channels=1, max_samples=36000, layout=B.LAYOUT_CHANNEL_MAJOR, device=-1, kinds=0, flags=0, demod_flags=0, **params):
        self._lib = _lib()
        cfg = RxConfig()
        rc = self._lib.tetra_rx_default_config(C.byref(cfg))
        if rc:
            raise TetraDemodError(rc, "tetra_rx_default_config")
        cfg.demod.n_channels, cfg.demod.max_samples, cfg.demod.layout, cfg.demod.device = n_channels, max_samples, layout, device
        cfg.demod.flags = demod_flags
        for k, v in params.items():
            if k not in B.PARAMS:
                raise TypeError("unknown parameter %r" % k)
            setattr(cfg.demod, k, v)
        cfg.kinds, cfg.flags = kinds, flags
        self.n_channels, self.max_samples = n_channels, max_samples
        h = C.c_void_p()
        rc = self._lib.tetra_rx_create(C.byref(cfg), C.byref(h))
        if rc:
            raise TetraDemodError(rc, "tetra_rx_create")
        self._h = h
        self.max_rows = int(self._lib.tetra_rx_max_rows(h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tetra_rx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            raise TetraDemodError(rc, what)

    def reset(self):
        self._chk(self._lib.tetra_rx_reset(self._h), "tetra_rx_reset")

    def process(self, iq):
        iq = np.ascontiguousarray(iq, np.complex64)
        n = iq.size // self.n_channels
        self._chk(self._lib.tetra_rx_process(self._h, iq.ctypes.data_as(C.c_void_p), int(n)), "tetra_rx_process")

    def process_device(self, d_iq, n_samples, stream=None):
        p = C.c_void_p(d_iq.data_ptr() if hasattr(d_iq, "data_ptr") else int(d_iq))
        self._chk(self._lib.tetra_rx_process_device(self._h, p, int(n_samples), _stream_ptr(stream)), "tetra_rx_process_device")

    def wait(self):
        self._chk(self._lib.tetra_rx_wait(self._h), "tetra_rx_wait")

    def count(self, kind, which=0):
        n = C.c_int(0)
        self._chk(self._lib.tetra_rx_fetch(self._h, which, kind, None, None, 0, 0, C.byref(n)), "tetra_rx_fetch")
        return n.value

    def fetch(self, kind, which=0):
        """-> (blocks: structured array [n] of BLOCK_DTYPE, type1: uint8 [n][type1_bits(kind)])"""
        n = self.count(kind, which)
        nb = type1_bits(kind)
        blocks = np.zeros(max(n, 1), BLOCK_DTYPE)
        t1 = np.zeros((max(n, 1), nb), np.uint8)
        got = C.c_int(0)
        self._chk(self._lib.tetra_rx_fetch(self._h, which, kind, blocks.ctypes.data_as(C.c_void_p), t1.ctypes.data_as(C.c_void_p), nb,
                                           max(n, 1), C.byref(got)), "tetra_rx_fetch")
        return blocks[:got.value], t1[:got.value]

    def cells(self, first=0, count=None):
        count = self.n_channels - first if count is None else count
        arr = (CellState * max(count, 1))()
        self._chk(self._lib.tetra_rx_get_cell(self._h, first, count, arr), "tetra_rx_get_cell")
        return [arr[i] for i in range(count)]

    def sync_states(self, first=0, count=None):
        count = self.n_channels - first if count is None else count
        arr = (BsyncState * max(count, 1))()
        self._chk(self._lib.tetra_rx_get_sync_state(self._h, first, count, arr), "tetra_rx_get_sync_state")
        return [(arr[i].state, arr[i].bits_in_buf, arr[i].bitbuf_start_bitnum, arr[i].next_frame_start_bitnum) for i in range(count)]

    def stage_ms(self):
        ms = (C.c_float * 4)()
        self._chk(self._lib.tetra_rx_stage_ms(self._h, C.byref(ms)), "tetra_rx_stage_ms")
        return [float(v) for v in ms]

    def rows_device(self, kind, which=0, stream=None):
        """-> (d_type2 pointer, type2_stride, d_blocks pointer, d_n_rows pointer): raw device addresses."""
        t2, blk, nr, st = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
        self._chk(self._lib.tetra_rx_rows_device(self._h, which, kind, C.byref(t2), C.byref(st), C.byref(blk), C.byref(nr), _stream_ptr(stream)),
                  "tetra_rx_rows_device")
        return t2.value, st.value, blk.value, nr.value

    def bits_device(self, which=0, stream=None):
        bits, nb, st = C.c_void_p(), C.c_void_p(), C.c_int(0)
        self._chk(self._lib.tetra_rx_bits_device(self._h, which, C.byref(bits), C.byref(st), C.byref(nb), _stream_ptr(stream)),
                  "tetra_rx_bits_device")
        return bits.value, st.value, nb.value

    def demod_handle(self):
        return self._lib.tetra_rx_demod(self._h)
