"""Builds and binds tests/emul/emul.cpp (host emulation of the kernels' lane-level code)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libtetra_emul.so")
DEPS = [os.path.join(HERE, "emul.cpp"),
        os.path.join(ROOT, "sdrpp-tetra-demodulator_amd", "csrc", "demod_core.hpp"),
        os.path.join(ROOT, "sdrpp-tetra-demodulator_amd", "csrc", "design.hpp"),
        os.path.join(ROOT, "sdrpp-tetra-demodulator_amd", "csrc", "constellation_core.hpp"),
        os.path.join(ROOT, "include", "tetra_demod.h")]

sys.path.insert(0, ROOT)
import tetra_amd  # noqa: E402

Config = tetra_amd.pkg.binding.Config
ChannelState = tetra_amd.pkg.binding.ChannelState


class K1Consts(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("agc_set_point", "agc_rate", "agc_max_gain", "fll_alpha", "fll_beta",
                                         "fll_min_freq", "fll_max_freq")]


class K2Consts(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("tr_alpha", "tr_beta", "tr_min_freq", "tr_max_freq", "costas_alpha",
                                         "costas_beta", "costas_min_freq", "costas_max_freq")]


class Tables(C.Structure):
    _fields_ = [("ntaps", C.c_int), ("rrc", C.c_float * 144), ("be_re", C.c_float * 144), ("be_im", C.c_float * 144),
                ("bank", C.c_float * 1024), ("k1", K1Consts), ("k2", K2Consts), ("tr_omega", C.c_float)]


def build(force=False):
    stale = force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS)
    if stale:
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-mavx2", "-DTETRA_HOST_EMUL",
                        "-shared", "-fPIC", "-o", LIB, os.path.join(HERE, "emul.cpp")], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        vp = C.c_void_p
        L.emul_design.argtypes = [C.POINTER(Config), C.POINTER(Tables)]
        L.emul_design.restype = C.c_int
        L.emul_default_cfg.argtypes = [C.POINTER(Config)]
        L.emul_default_cfg.restype = None
        L.emul_reset_state.argtypes = [C.POINTER(Tables), C.POINTER(ChannelState)]
        L.emul_reset_state.restype = None
        L.emul_fused.argtypes = [C.POINTER(Tables), C.POINTER(ChannelState), C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp]
        L.emul_fused.restype = C.c_int
        L.emul_fused_shape.argtypes = L.emul_fused.argtypes + [C.c_int]
        L.emul_fused_shape.restype = C.c_int
        L.emul_bits_stride_for.argtypes = [C.POINTER(Config), C.c_longlong]
        L.emul_bits_stride_for.restype = C.c_longlong
        L.emul_quality_distance.argtypes = [C.c_int, vp, vp]
        L.emul_quality_distance.restype = None
        L.emul_constellation.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.c_int]
        L.emul_constellation.restype = None
        _lib = L
    return _lib


def default_cfg():
    cfg = Config()
    lib().emul_default_cfg(C.byref(cfg))
    return cfg


def design(cfg=None):
    cfg = cfg if cfg is not None else default_cfg()
    t = Tables()
    rc = lib().emul_design(C.byref(cfg), C.byref(t))
    if rc:
        raise ValueError("emul_design failed")
    return t


def bits_stride_for(cfg, n):
    """The product's row rule for a configuration (-1 = the product refuses the parameters)."""
    return int(lib().emul_bits_stride_for(C.byref(cfg), int(n)))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def quality_distance(z):
    """k_quality's per-symbol distance (demod_core.hpp quality_distance) for complex64 symbols z[n]."""
    import numpy as np
    z = np.ascontiguousarray(z, np.complex64)
    out = np.zeros(z.size, np.float32)
    lib().emul_quality_distance(int(z.size), _p(z), _p(out))
    return out


class ConstellationTap:
    """One channel of k_constellation (constellation_core.hpp) on the host: feed(z) = one call's symbols."""

    def __init__(self, nthr=256):
        import numpy as np
        self.blk = np.zeros(1024, np.complex64)
        self.part = np.full(1024, np.nan + 1j * np.nan, np.complex64)      # never read beyond `fill`: NaNs would show
        self.fill = np.zeros(1, np.int32)
        self.blocks = np.zeros(1, np.int32)
        self.nthr = nthr

    def feed(self, z):
        import numpy as np
        z = np.ascontiguousarray(z, np.complex64)
        lib().emul_constellation(int(z.size), _p(z), _p(self.blk), _p(self.part), _p(self.fill), _p(self.blocks), self.nthr)


def bits_stride(n):
    s = int(n / 0.95) + 16
    return (s + 15) // 16 * 16


class EmulDemod:
    """C <= 64 channels through the emulated kernels, with carried state."""

    def __init__(self, n_channels=1, cfg=None, fused=True, fll_lanes=8):
        """fll_lanes: 8 = the FLL rows of the 16-channel workgroup, 4 = those of the 32-channel workgroup."""
        self.fused = True
        self.fll_lanes = fll_lanes
        self.C = n_channels
        self.tab = design(cfg)
        self.st = (ChannelState * n_channels)()
        for c in range(n_channels):
            lib().emul_reset_state(C.byref(self.tab), C.byref(self.st[c]))

    def process(self, iq, want_sym=False):
        iq = np.ascontiguousarray(iq, np.complex64)
        if iq.ndim == 1:
            iq = iq[None, :]
        Cn, n = iq.shape
        assert Cn == self.C
        y = np.zeros((Cn, n), np.complex64)
        stride = bits_stride(n)
        bits = np.zeros((Cn, stride), np.uint8)
        nb = np.zeros(Cn, np.int32)
        sym = np.zeros((Cn, stride // 2), np.complex64) if want_sym else None
        rc = lib().emul_fused_shape(C.byref(self.tab), self.st, Cn, n, _p(iq), _p(y), _p(bits), stride, _p(nb), _p(sym), self.fll_lanes)
        assert rc == 0, rc
        return dict(y=y, bits=bits, n_bits=nb, sym=sym)
