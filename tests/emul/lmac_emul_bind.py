"""Builds and binds tests/emul/lmac_emul.cpp (host build of the lower-MAC decoder's lane-level code)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "liblmac_emul.so")
DEPS = [os.path.join(HERE, "lmac_emul.cpp"), os.path.join(ROOT, "sdrpp-tetra-demodulator_amd", "csrc", "lmac_core.hpp"),
        os.path.join(ROOT, "sdrpp-tetra-demodulator_amd", "csrc", "demux_core.hpp")]

# tetra_blk_param[] values (include/tetra_lmac.h; the product's copy is checked against these in tests/test_abi.py)
BLK_PARAM = {0: (120, 80, 60, 11), 1: (216, 144, 124, 101), 2: (216, 144, 124, 101), 4: (168, 112, 92, 13), 5: (432, 288, 268, 103)}

_lib = None


def build(force=False):
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS)
    if stale:
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", LIB, os.path.join(HERE, "lmac_emul.cpp")], check=True)
    return LIB


def decode_batch(blk_type, type5, scramb):
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
    n345, n2, n1, a = BLK_PARAM[blk_type]
    rows = np.ascontiguousarray(type5, np.uint8)
    n, stride = rows.shape
    si = np.ascontiguousarray(scramb, np.uint32)
    out = np.zeros((n, n2), np.uint8)
    ok = np.zeros(n, np.int32)
    vp = C.c_void_p
    rc = _lib.lmac_emul_decode(n345, n2, n1, a, rows.ctypes.data_as(vp), n, stride, si.ctypes.data_as(vp), out.ctypes.data_as(vp), n2,
                               ok.ctypes.data_as(vp))
    assert rc == 0
    return out, ok


def decode_frames(tpsap, blk_num, frames_packed, frame_type, row_frame, frame_scramb, out_stride):
    """The lane code of k_lmac_frames (tetra_lmac_decode_frames_device, one job) for the listed frames: (rows [n][out_stride], crc_ok)."""
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
    fr = np.ascontiguousarray(frames_packed, np.uint32)
    ft = np.ascontiguousarray(frame_type, np.int32)
    rf = np.ascontiguousarray(row_frame, np.int32)
    sc = None if frame_scramb is None else np.ascontiguousarray(frame_scramb, np.uint32)
    out = np.zeros((rf.size, out_stride), np.uint8)
    ok = np.zeros(rf.size, np.int32)
    vp = C.c_void_p
    rc = _lib.lmac_emul_decode_frames(int(tpsap), int(blk_num), fr.ctypes.data_as(vp), ft.ctypes.data_as(vp), rf.ctypes.data_as(vp), rf.size,
                                      None if sc is None else sc.ctypes.data_as(vp), out.ctypes.data_as(vp), out_stride, ok.ctypes.data_as(vp))
    if rc:
        raise ValueError("refused")
    return out, ok
