"""Builds and binds tests/emul/bsync_emul.cpp (host build of the burst synchroniser's logic, csrc/bsync_core.hpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libbsync_emul.so")
DEPS = [os.path.join(HERE, "bsync_emul.cpp"), os.path.join(ROOT, "sdrpp-tetra-demodulator_amd", "csrc", "bsync_core.hpp")]

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
