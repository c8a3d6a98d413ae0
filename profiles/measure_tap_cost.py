"""Cost of the post-launch taps on the headline workload: k_fused alone vs + k_constellation vs + k_quality (HIP events of the handle)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetra_amd
pkg = tetra_amd.pkg; B = pkg.binding
C, N = 4096, 36000
dev = torch.device("cuda", 0)
base = np.stack([pkg.synth.gen_channel(N, 100 + c)[0] for c in range(64)])
d_iq = torch.from_numpy(np.tile(base, (C // 64, 1))).to(dev)
stride = B.bits_stride(N)
d_bits = torch.zeros((C, stride), dtype=torch.uint8, device=dev); d_nb = torch.zeros(C, dtype=torch.int32, device=dev)
out = {}
for name, flags in (("plain", 0), ("constellation", B.FLAG_CONSTELLATION), ("quality", B.FLAG_QUALITY), ("both", B.FLAG_CONSTELLATION | B.FLAG_QUALITY), ("plain_again", 0)):
    d = pkg.Demodulator(C, N, flags=flags)
    s = torch.cuda.current_stream(dev)
    for _ in range(12): d.process_device(d_iq, N, d_bits, stride, d_nb, stream=s)
    torch.cuda.synchronize()
    out[name] = round(float(np.mean(d.kernel_ms_history(6))), 4)
    d.close()
print(json.dumps(out))
