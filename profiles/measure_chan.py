"""Times the channeliser kernel of BASELINE config 5 (800 channels, 8 taps per channel, D = 400; 5e6 wideband samples -> 12500
frames) in its three forms, alternating on steady clocks: the 32 x 5 x 5 mixed-radix FFT (default at D = M / 2 since round 5), the
matrix-pipe DFT (round 4's form, TETRA_CHAN_FLAG_MATRIX_DFT) and the direct-sum kernel (TETRA_CHAN_FLAG_VALU_DFT), HIP events on the
launch stream; checks that the outputs agree to the float32 tolerance."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetra_amd  # noqa: E402

pkg = tetra_amd.pkg
dev = torch.device("cuda", 0)
M, P, D, n_in = 800, 8, 400, 5000000
frames = n_in // D
g = torch.Generator(device=dev)
g.manual_seed(3)
x = torch.view_as_complex(torch.randn((n_in, 2), device=dev, generator=g)).contiguous()
s = torch.cuda.current_stream(dev)
chs = {"fft": pkg.Channeliser(M, P, D, max_in=n_in), "mfma": pkg.Channeliser(M, P, D, max_in=n_in, flags=pkg.Channeliser.FLAG_MATRIX_DFT),
       "valu": pkg.Channeliser(M, P, D, max_in=n_in, flags=pkg.Channeliser.FLAG_VALU_DFT)}
outs = {k: torch.zeros((frames, M), dtype=torch.complex64, device=dev) for k in chs}
for _ in range(20):
    for k, ch in chs.items():
        ch.process_device(x, n_in, outs[k], s)
torch.cuda.synchronize()
ms = {k: [] for k in chs}
for _ in range(20):
    for k, ch in chs.items():
        ch.process_device(x, n_in, outs[k], s)
        torch.cuda.synchronize()
        ms[k].append(ch.last_kernel_ms())
# same stream position on both handles (every call consumed n_in samples): outputs comparable frame by frame
diff = float((outs["mfma"] - outs["valu"]).abs().max() / outs["valu"].abs().max())
diff_fft = float((outs["fft"] - outs["valu"]).abs().max() / outs["valu"].abs().max())
by = 8.0 * n_in + 8.0 * frames * M
res = {"workload": "%d samples -> %d frames x %d channels" % (n_in, frames, M), "max_rel_difference_mfma_vs_valu": diff,
       "max_rel_difference_fft_vs_valu": diff_fft}
import math
for k in chs:
    fl = frames * (4.0 * P * M + (5.0 * M * math.log2(M) if k == "fft" else 8.0 * M * 57))
    t = sorted(ms[k])[len(ms[k]) // 2]
    res[k] = {"kernel_ms_median": round(t, 4), "GBps": round(by / (t * 1e-3) / 1e9, 1), "frac_hbm_8TBps": round(by / (t * 1e-3) / 8e12, 4), "frac_hbm_achievable_6.29TBps": round(by / (t * 1e-3) / 6.29e12, 4),
              "TFLOPs_algorithmic": round(fl / (t * 1e-3) / 1e12, 2), "frac_fp32_157": round(fl / (t * 1e-3) / 157.3e12, 4)}
print(json.dumps(res))
