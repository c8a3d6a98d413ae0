import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tetra_amd, bench
pkg = tetra_amd.pkg
dev = torch.device("cuda", 0)
C, N = 4096, 36000
iq, _ = bench.make_input(torch, pkg.synth, dev, C, N, seed=20260000)
stride = pkg.binding.bits_stride(N)
bits = torch.zeros((C, stride), dtype=torch.uint8, device=dev); nb = torch.zeros(C, dtype=torch.int32, device=dev)
dem = pkg.Demodulator(C, N)
st = torch.cuda.current_stream(dev)
def run(k):
    for _ in range(k): dem.process_device(iq, N, bits, stride, nb, None, st)
    torch.cuda.synchronize()
    return [round(float(x), 3) for x in dem.kernel_ms_history(k)]
print("fresh     ", run(12))
dem.reset(); print("after reset", run(6))
time.sleep(1.0); print("after 1s idle", run(6))
dem.reset(); time.sleep(1.0); print("reset+idle", run(6))
