import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch, tetra_amd
pkg = tetra_amd.pkg
dev = torch.device('cuda', 0)
for C, N, prm in ((4096, 36000, dict(rrc_tap_count=100)), (4096, 36000, dict(samplerate=18000.0 * 0.2)), (256, 36000, dict(rrc_tap_count=129))):
    d = pkg.Demodulator(C, N, **prm)
    iq = torch.view_as_complex(torch.randn((C, N, 2), device=dev) * 0.3).contiguous()
    stride = d.bits_stride(N)
    bits = torch.zeros((C, stride), dtype=torch.uint8, device=dev); nb = torch.zeros(C, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream(dev)
    d.process_device(iq, N, bits, stride, nb, None, s); torch.cuda.synchronize()
    t = time.time(); d.process_device(iq, N, bits, stride, nb, None, s); torch.cuda.synchronize()
    print(C, N, prm, 'ms', round((time.time() - t) * 1e3, 1), 'kernel_ms', round(float(d.kernel_ms_history(1)[0]), 1), 'stride', stride)
    d.close()
