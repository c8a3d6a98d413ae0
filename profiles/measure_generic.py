"""Launch times of the parameter sets beyond the fused kernel's regular rows (one GPU, resident input, one warm-up launch):
filters of 73 .. 129 taps in the fused kernel's LONG rows and, with TETRA_FLAG_GENERIC_KERNEL, in the generic kernel; timing loops
below 0.27 samples per symbol (always the generic kernel).    gpurun -- 'python profiles/measure_generic.py'"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, tetra_amd
pkg = tetra_amd.pkg
B = pkg.binding
dev = torch.device('cuda', 0)
CASES = ((256, dict(rrc_tap_count=100), 0), (1024, dict(rrc_tap_count=100), 0), (1028, dict(rrc_tap_count=100), 0), (4096, dict(rrc_tap_count=100), 0),
         (4096, dict(rrc_tap_count=100), B.FLAG_SMALL_WORKGROUPS), (8192, dict(rrc_tap_count=100), 0),
         (4096, dict(rrc_tap_count=129), 0), (4096, dict(rrc_tap_count=65), 0),
         (4096, dict(rrc_tap_count=100), B.FLAG_GENERIC_KERNEL), (256, dict(rrc_tap_count=129), B.FLAG_GENERIC_KERNEL),
         (1024, dict(samplerate=18000.0 * 0.2), 0), (4096, dict(samplerate=18000.0 * 0.2), 0), (4096, dict(samplerate=18000.0 * 0.12), 0),
         (4096, dict(samplerate=18000.0 * 0.2), B.FLAG_GENERIC_KERNEL), (4096, dict(samplerate=18000.0 * 0.06), 0))
N = 36000
for C, prm, flags in CASES:
    d = pkg.Demodulator(C, N, flags=flags, **prm)
    iq = torch.view_as_complex(torch.randn((C, N, 2), device=dev) * 0.3).contiguous()
    stride = d.bits_stride(N)
    bits = torch.zeros((C, stride), dtype=torch.uint8, device=dev); nb = torch.zeros(C, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream(dev)
    for _ in range(2):
        d.process_device(iq, N, bits, stride, nb, None, s)
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        d.process_device(iq, N, bits, stride, nb, None, s); torch.cuda.synchronize()
        ms.append(float(d.kernel_ms_history(1)[0]))
    print(json.dumps(dict(channels=C, samples=N, params=prm, flags=flags, kernel="generic" if flags & B.FLAG_GENERIC_KERNEL or prm.get("samplerate", 36000.0) < 18000.0 * 0.07 else "fused, 1024-deep symbol ring" if "samplerate" in prm else ("fused, long rows" if prm.get("rrc_tap_count", 65) > 72 else "fused"),
                          kernel_ms=round(min(ms), 3), msamples_s=round(C * N / min(ms) / 1e3, 1))))
    d.close()
