#!/usr/bin/env python3
"""Latency of the single-channel drop-in call (BASELINE config 1's shape on the GPU path): tetra_demod_process with C = 1 and the
180-sample chunks SDR++ delivers at 36 ksps, host buffers in and out, symbols requested like PI4DQPSK::process does.
One JSON line: microseconds per call (ctypes overhead included) and the real-time duty cycle it means."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tetra_amd  # noqa: E402

pkg = tetra_amd.pkg
out = {}
for n in (180, 1024, 36000):
    iq, _, _ = pkg.synth.gen_channel(max(n, 36000), 5)
    d = pkg.Demodulator(1, 65536, flags=pkg.binding.FLAG_REFERENCE_QUIRKS)
    stride = d.bits_stride(n)
    bits = np.zeros((1, stride), np.uint8)
    nb = np.zeros(1, np.int32)
    sym = np.zeros((1, stride // 2), np.complex64)
    vp = C.c_void_p
    blk = np.ascontiguousarray(iq[None, :n])
    args = (d._h, blk.ctypes.data_as(vp), n, bits.ctypes.data_as(vp), stride, nb.ctypes.data_as(vp), sym.ctypes.data_as(vp))
    for _ in range(50):
        d._lib.tetra_demod_process(*args)
    reps = 2000 if n < 5000 else 200
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = d._lib.tetra_demod_process(*args)
    el = time.perf_counter() - t0
    assert rc == 0
    out["us_per_call_n%d" % n] = round(el / reps * 1e6, 1)
    out["kernel_us_n%d" % n] = round(d.last_kernel_ms() * 1e3, 1)
    out["realtime_duty_n%d" % n] = round(el / reps / (n / 36000.0), 4)
    d.close()
print(json.dumps(out))
