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
