"""A/B of k_channelise_fft variants (experiment switches in cfg.reserved) on config 5's geometry: alternating launches on steady
clocks, median of HIP-event times.  Usage: python profiles/measure_chan_fft.py [flags ...] (default: 0 and 0x100).  The switches
exist only in a library built with -DTETRA_CHAN_EXPERIMENTS (profiles/build_exp.sh); the product library refuses them."""
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
flags = [int(a, 0) for a in sys.argv[1:]] or [0, 0x100]
g = torch.Generator(device=dev)
g.manual_seed(3)
x = torch.view_as_complex(torch.randn((n_in, 2), device=dev, generator=g)).contiguous()
s = torch.cuda.current_stream(dev)
chs = {f: pkg.Channeliser(M, P, D, max_in=n_in, flags=f) for f in flags}
out = torch.zeros((frames, M), dtype=torch.complex64, device=dev)
for _ in range(30):
    for f, ch in chs.items():
        ch.process_device(x, n_in, out, s)
torch.cuda.synchronize()
ms = {f: [] for f in flags}
for _ in range(40):
    for f, ch in chs.items():
        ch.process_device(x, n_in, out, s)
        torch.cuda.synchronize()
        ms[f].append(ch.last_kernel_ms())
by = 8.0 * n_in + 8.0 * frames * M
res = {}
for f in flags:
    v = sorted(ms[f])
    t = v[len(v) // 2]
    res[hex(f)] = {"kernel_ms_median": round(t, 4), "min": round(v[0], 4), "GBps": round(by / (t * 1e-3) / 1e9, 1), "frac_of_6.29TBps": round(by / (t * 1e-3) / 6.29e12, 3)}
print(json.dumps(res))
