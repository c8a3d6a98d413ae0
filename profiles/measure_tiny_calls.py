#!/usr/bin/env python3
"""In-place small calls (tetra_demod_process reading / writing page-locked host blocks from the kernels) against the copy-engine
form, for a few call shapes: microseconds per call of each, and that both deliver the same bits and symbols.  Needs an
experiment build with -DTETRA_EXP_TINY_ENV (profiles/build_exp.sh tiny "-DTETRA_EXP_TINY_ENV"); one process per mode because
the limit is read once:   python profiles/measure_tiny_calls.py            (spawns itself for both modes)"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(1, 180), (1, 1024), (4, 180), (16, 180), (64, 180), (16, 1024), (64, 500)]


def one_mode():
    import ctypes as C
    import numpy as np
    sys.path.insert(0, ROOT)
    import tetra_amd
    pkg = tetra_amd.pkg
    out = {}
    for (Cn, n) in SHAPES:
        iq, _, _ = pkg.synth.gen_batch(Cn, 8 * n, base_seed=77)
        d = pkg.Demodulator(Cn, 65536, flags=pkg.binding.FLAG_REFERENCE_QUIRKS)
        stride = d.bits_stride(n)
        bits = np.zeros((Cn, stride), np.uint8)
        nb = np.zeros(Cn, np.int32)
        sym = np.zeros((Cn, stride // 2), np.complex64)
        vp = C.c_void_p
        h = hashlib.sha256()
        for k in range(8):          # eight consecutive calls with carried state: the digest of everything delivered
            blk = np.ascontiguousarray(iq[:, k * n:(k + 1) * n])
            rc = d._lib.tetra_demod_process(d._h, blk.ctypes.data_as(vp), n, bits.ctypes.data_as(vp), stride, nb.ctypes.data_as(vp),
                                            sym.ctypes.data_as(vp))
            assert rc == 0
            for c in range(Cn):
                h.update(bits[c, :nb[c]].tobytes())
                h.update(sym[c, :nb[c] // 2].tobytes())
        blk = np.ascontiguousarray(iq[:, :n])
        args = (d._h, blk.ctypes.data_as(vp), n, bits.ctypes.data_as(vp), stride, nb.ctypes.data_as(vp), sym.ctypes.data_as(vp))
        for _ in range(100):
            d._lib.tetra_demod_process(*args)
        reps = 1500
        t0 = time.perf_counter()
        for _ in range(reps):
            d._lib.tetra_demod_process(*args)
        el = time.perf_counter() - t0
        out["%dx%d" % (Cn, n)] = dict(us_per_call=round(el / reps * 1e6, 1), kernel_us=round(d.last_kernel_ms() * 1e3, 1),
                                      digest=h.hexdigest()[:16])
        d.close()
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one_mode()
    else:
        res = {}
        for name, limit in (("copy_engine", "0"), ("in_place", "100000")):
            env = dict(os.environ, TETRA_TINY_SAMPLES=limit)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True, check=True)
            res[name] = json.loads(r.stdout.strip().splitlines()[-1])
        same = all(res["copy_engine"][k]["digest"] == res["in_place"][k]["digest"] for k in res["copy_engine"])
        print(json.dumps(dict(same_outputs=same, **res)))
