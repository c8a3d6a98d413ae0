"""Generates tests/golden/lmac_golden.npz: lower-MAC input rows (type-5 bits, scrambling codes) and the outputs of the
REFERENCE's own primitives on them (oracle/_ref/libtetra_lmac_ref.so, built from /root/reference by oracle/build_ref.sh and
chained like tp_sap_udata_ind, src/decoder/src/lower_mac/tetra_lower_mac.c:181-236).  Data only; run from the repo root in
the build container:  python tests/golden/make_lmac_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_binding as R  # noqa: E402
from tests.test_lmac import make_rows, ref_decode_rows  # noqa: E402

out = {}
for t in (0, 1, 2, 4, 5):
    rows, si, sent = make_rows(R, t, 48, 7000 + t)
    type2, ok = ref_decode_rows(R, t, rows, si)
    out[f"rows_{t}"], out[f"scramb_{t}"], out[f"type2_{t}"], out[f"crc_ok_{t}"] = rows, si, type2, ok
rng = np.random.default_rng(7003)
rows = rng.integers(0, 2, (48, 32), dtype=np.uint8)
si = rng.integers(0, 2 ** 32, 48, dtype=np.uint64).astype(np.uint32)
out["rows_3"], out["scramb_3"] = rows, si
out["type2_3"] = np.stack([R.lmac_decode(3, rows[b], si[b])[0] for b in range(48)])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lmac_golden.npz"), **out)
print("wrote lmac_golden.npz", {k: v.shape for k, v in out.items()})
