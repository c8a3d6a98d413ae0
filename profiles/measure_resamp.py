"""Times the 18 / 25 resampler kernel of BASELINE config 5 (800 channels: 12500 frames at 50 ksps -> 9000 at 36 ksps) in its variants,
alternating on steady clocks, HIP events on the launch stream: 16-byte and 8-byte lane units, 8 / 12 / 16 / 24 taps per phase, the
generic kernel; checks that the variants agree to the float32 tolerance.  `python profiles/measure_resamp.py [variant]` runs one
variant only (for the counter passes)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetra_amd  # noqa: E402

pkg = tetra_amd.pkg
dev = torch.device("cuda", 0)
M, frames = 800, 12500
n36 = frames * 18 // 25
g = torch.Generator(device=dev)
g.manual_seed(3)
x = torch.view_as_complex(torch.randn((frames, M, 2), device=dev, generator=g)).contiguous()
s = torch.cuda.current_stream(dev)
R = pkg.Resampler
variants = {"t16_w4": dict(taps_per_phase=16), "t16_w2": dict(taps_per_phase=16, flags=R.FLAG_NARROW_UNITS),
            "t8_w4": dict(taps_per_phase=8), "t12_w4": dict(taps_per_phase=12), "t24_w4": dict(taps_per_phase=24),
            "t24_w2": dict(taps_per_phase=24, flags=R.FLAG_NARROW_UNITS), "t16_generic": dict(taps_per_phase=16, flags=R.FLAG_GENERIC)}
only = sys.argv[1] if len(sys.argv) > 1 else None
if only:
    variants = {only: variants[only]}
rs = {k: R(M, 18, 25, max_in=frames, **kw) for k, kw in variants.items()}
outs = {k: torch.zeros((n36 + 1, M), dtype=torch.complex64, device=dev) for k in rs}
for _ in range(20):
    for k, r in rs.items():
        assert r.process_device(x, frames, outs[k], s) == n36
torch.cuda.synchronize()
ms = {k: [] for k in rs}
for _ in range(20):
    for k, r in rs.items():
        r.process_device(x, frames, outs[k], s)
        torch.cuda.synchronize()
        ms[k].append(r.last_kernel_ms())
by = 8.0 * frames * M + 8.0 * n36 * M
res = {"workload": "%d frames x %d channels @ 50 ksps -> %d frames @ 36 ksps" % (frames, M, n36), "algorithmic_bytes": by}
if "t16_w4" in outs:
    for k in ("t16_w2", "t16_generic"):
        if k in outs:
            res["max_rel_difference_%s_vs_t16_w4" % k] = float((outs[k] - outs["t16_w4"]).abs().max() / outs["t16_w4"].abs().max())
for k in rs:
    t = sorted(ms[k])[len(ms[k]) // 2]
    res[k] = {"kernel_ms_median": round(t, 4), "GBps": round(by / (t * 1e-3) / 1e9, 1), "frac_hbm_8TBps": round(by / (t * 1e-3) / 8e12, 4),
              "frac_hbm_achievable_6.29TBps": round(by / (t * 1e-3) / 6.29e12, 4)}
print(json.dumps(res))
