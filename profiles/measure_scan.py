"""Times k_find_train_seq on demodulator-sized rows (4096 channels x 36864 bits) with HIP events (torch.cuda.Event on the
stream the kernel is launched on) and prints achieved HBM GB/s: algorithmic bytes = one byte per scanned position."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetra_amd  # noqa: E402

pkg = tetra_amd.pkg
dev = torch.device("cuda", 0)
C, n = 4096, 36864
stride = n + 64
g = torch.Generator(device=dev)
g.manual_seed(1)
bits = torch.randint(0, 2, (C, stride), dtype=torch.uint8, device=dev, generator=g)
end = torch.full((C,), n, dtype=torch.int32, device=dev)
t = torch.zeros(C, dtype=torch.int32, device=dev)
o = torch.zeros(C, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream(dev)
# random bits: a 22-bit head matches by chance ~ 5 * 36864 / 4M = 4 % of rows; mask 0 -> nothing verifies -> full scan of every row
for mask, name in ((0x00, "full scan (no sequence enabled: every row is read to the end)"), (0x1f, "all sequences, random data")):
    for _ in range(3):
        pkg.scan_binding.find_train_seq_batch_device(bits, C, stride, end, mask, t, o, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    reps = 20
    for _ in range(reps):
        pkg.scan_binding.find_train_seq_batch_device(bits, C, stride, end, mask, t, o, s)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"kernel": "k_find_train_seq", "case": name, "channels": C, "bits_per_channel": n, "ms": round(ms, 4),
                      "achieved_GBps": round(C * n / ms / 1e6, 1), "frac_of_8TBps": round(C * n / ms / 1e6 / 8000, 4),
                      "found": int((t >= 0).sum())}))
