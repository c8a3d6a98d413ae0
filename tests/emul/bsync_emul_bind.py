"""Builds and binds tests/emul/bsync_emul.cpp (host build of the burst synchroniser's logic, csrc/bsync_core.hpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libbsync_emul.so")
DEPS = [os.path.join(HERE, "bsync_emul.cpp"), os.path.join(ROOT, "sdrpp-tetra-demodulator_amd", "csrc", "bsync_core.hpp"),
        os.path.join(ROOT, "sdrpp-tetra-demodulator_amd", "csrc", "demux_core.hpp"), os.path.join(ROOT, "include", "tetra_burst_sync.h")]

_lib = None


def build(force=False):
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS)
    if stale:
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", LIB, os.path.join(HERE, "bsync_emul.cpp")], check=True)
    return LIB


class Emul:
    """One channel: same interface as oracle.binding.BurstSyncOracle."""

    def __init__(self, batch=True):
        """batch: LOCKED steady state evaluated frame-parallel like the kernel does (run()'s batch hook) or event by event."""
        self.batch = 1 if batch else 0
        global _lib
        if _lib is None:
            build()
            _lib = C.CDLL(LIB)
            _lib.bsync_emul_process.restype = C.c_int
        self.st = np.zeros(4, np.uint32)
        self.carry = np.zeros(4096, np.uint8)

    @property
    def state(self):
        return int(self.st[0]), int(self.st[1]), int(self.st[2]), int(self.st[3])

    def feed(self, bits):
        b = np.ascontiguousarray(bits, np.uint8)
        cap = (4096 + b.size) // 510 + 4
        fr = np.zeros((cap, 512), np.uint8)
        ty = np.zeros(cap, np.int32)
        bn = np.zeros(cap, np.uint32)
        vp = C.c_void_p
        n = _lib.bsync_emul_process(self.st.ctypes.data_as(vp), self.carry.ctypes.data_as(vp), b.ctypes.data_as(vp), b.size,
                                    fr.ctypes.data_as(vp), ty.ctypes.data_as(vp), bn.ctypes.data_as(vp), cap, self.batch)
        assert n >= 0
        return fr[:n, :510].copy(), ty[:n].copy(), bn[:n].copy()


def _lib_demux():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.bsync_emul_process.restype = C.c_int
    return _lib


def pack_frames(frames):
    """[n][512] bytes (510 bits + 2 zero bytes) -> [n][16] uint32, first bit = most significant bit of word 0 (k_burst_sync<true>'s form)."""
    f = np.ascontiguousarray(frames, np.uint8) & 1
    return np.ascontiguousarray(np.packbits(f, axis=1).view(">u4").astype(np.uint32))


def demux(frames, frame_type, tpsap, blk_num, row_stride, packed=False, fill=9):
    """The thread-level code of tetra_burst_demux[_packed]_device run for every thread of its launch: (rows [n][row_stride], valid [n])."""
    ft = np.ascontiguousarray(frame_type, np.int32)
    n = ft.size
    src = pack_frames(frames) if packed else np.ascontiguousarray(frames, np.uint8)
    rows = np.full((n, row_stride), fill, np.uint8)
    valid = np.full(n, fill, np.int32)
    vp = C.c_void_p
    rc = _lib_demux().bsync_emul_demux(src.ctypes.data_as(vp), int(packed), ft.ctypes.data_as(vp), n, tpsap, blk_num, rows.ctypes.data_as(vp),
                                       row_stride, valid.ctypes.data_as(vp))
    if rc:
        raise ValueError("refused")
    return rows, valid


def demux_compact(frames, frame_type, tpsap, blk_num, row_stride, packed=False, fill=9):
    """tetra_burst_demux_compact[_packed]_device: (rows [n][row_stride], row_frame [n], n_rows)."""
    ft = np.ascontiguousarray(frame_type, np.int32)
    n = ft.size
    src = pack_frames(frames) if packed else np.ascontiguousarray(frames, np.uint8)
    rows = np.full((n, row_stride), fill, np.uint8)
    row_frame = np.full(n, -1, np.int32)
    cnt = np.zeros(1, np.int32)
    vp = C.c_void_p
    rc = _lib_demux().bsync_emul_demux_compact(src.ctypes.data_as(vp), int(packed), ft.ctypes.data_as(vp), n, tpsap, blk_num,
                                               rows.ctypes.data_as(vp), row_stride, row_frame.ctypes.data_as(vp), cnt.ctypes.data_as(vp))
    if rc:
        raise ValueError("refused")
    return rows, row_frame, int(cnt[0])
