"""Generates the frozen fixtures under tests/golden/ from the CPU oracle.

NOT reference-derived: the reference has no vectors and cannot be built here (see oracle/tetra_oracle.h);
these pin THIS restatement's arithmetic so later edits cannot drift silently.  Inputs come from the
package's synthetic generator with fixed seeds.  Run from the repo root: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tetra_amd  # noqa: E402
from oracle import binding as ob  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    synth = tetra_amd.pkg.synth
    Cn, N = 3, 6000
    iq, _, _ = synth.gen_batch(Cn, N, base_seed=600)
    stride = ob.bits_stride(N)
    bits = np.zeros((Cn, stride), np.uint8)
    sym = np.zeros((Cn, stride // 2), np.complex64)
    nb = np.zeros(Cn, np.int32)
    ff = np.zeros(Cn, np.float32)
    om = np.zeros(Cn, np.float32)
    for c in range(Cn):
        o = ob.Oracle()
        r = o.process(iq[c])
        nb[c] = r["bits"].size
        bits[c, : nb[c]] = r["bits"]
        sym[c, : nb[c] // 2] = r["sym"]
        ff[c] = o.st.fll_freq
        om[c] = o.st.omega
    np.savez_compressed(os.path.join(HERE, "golden_c3_n6000.npz"), iq=iq, bits=bits, sym=sym, n_bits=nb,
                        fll_freq=ff, omega=om)
    # ETSI EN 300 392-2 clause 9.4.4.3.2-4 training sequences (protocol constants; the reference holds the
    # same bits at src/decoder/src/phy/tetra_burst.c:61-72 and src/main.cpp:457-468).
    ts = {
        "source": "ETSI EN 300 392-2 9.4.4.3.2 / 9.4.4.3.3 / 9.4.4.3.4",
        "normal_1": [1, 1, 0, 1, 0, 0, 0, 0, 1, 1, 1, 0, 1, 0, 0, 1, 1, 1, 0, 1, 0, 0],
        "normal_2": [0, 1, 1, 1, 1, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 1, 1, 1, 1, 0],
        "normal_3": [1, 0, 1, 1, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 1, 0, 1, 0, 1, 1, 0, 1],
        "extended": [1, 0, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, 1, 1, 1, 0, 1, 0, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, 1, 1],
        "sync": [1, 1, 0, 0, 0, 0, 0, 1, 1, 0, 0, 1, 1, 1, 0, 0, 1, 1, 1, 0, 1, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 1, 1,
                 0, 0, 1, 1, 1],
    }
    with open(os.path.join(HERE, "etsi_training_sequences.json"), "w") as f:
        json.dump(ts, f, indent=1)


if __name__ == "__main__":
    main()
