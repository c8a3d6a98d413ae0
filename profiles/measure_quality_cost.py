#!/usr/bin/env python3
"""What TETRA_FLAG_QUALITY (DQPSKSymbolExtractor's standarderr / sync statistic) and the optional symbol output cost at the
bench workload (4096 channels x 36000 samples, inputs resident): HIP-event time per call, mean of the last 8 of 16 calls.

    python profiles/measure_quality_cost.py            # prints one JSON line
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import tetra_amd  # noqa: E402

pkg = tetra_amd.pkg
dev = torch.device("cuda", 0)
C, N = 4096, 36000
iq, _ = bench.make_input(torch, pkg.synth, dev, C, N, seed=20260000)
stride = pkg.binding.bits_stride(N)
bits = torch.zeros((C, stride), dtype=torch.uint8, device=dev)
nb = torch.zeros(C, dtype=torch.int32, device=dev)
sym = torch.zeros((C, stride // 2, 2), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream(dev)
out = {"workload": f"{C}x{N}", "unit": "ms per call"}
for name, fl, s in (("plain", 0, None), ("plain+symbols", 0, sym), ("quality", pkg.binding.FLAG_QUALITY, None),
                    ("quality+symbols", pkg.binding.FLAG_QUALITY, sym)):
    dem = pkg.Demodulator(C, N, flags=fl)
    for _ in range(8):
        dem.process_device(iq, N, bits, stride, nb, s, st)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    ev[0].record(st)
    for i in range(8):
        dem.process_device(iq, N, bits, stride, nb, s, st)
        ev[i + 1].record(st)
    torch.cuda.synchronize()
    out[name] = round(ev[0].elapsed_time(ev[8]) / 8, 4)
    dem.close()
print(json.dumps(out))
