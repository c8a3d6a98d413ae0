"""Times k_lmac_decode with HIP events (torch.cuda.Event on the stream the kernel is launched on).  Workload: one second of
signal on 4096 channels = 70.6 slots per channel; per slot one SCH/F block (432 type-5 bits) -> 4096 * 70 = 286720 blocks
per launch, plus the same count as two half-slot NDB blocks (216 bits) each.  Reports blocks/s, decoded (type-2) Mbit/s and
the trellis rate (add-compare-select butterflies per second = blocks * (type2 + 4) steps * 8)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetra_amd  # noqa: E402

pkg = tetra_amd.pkg
lb = pkg.lmac_binding
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
s = torch.cuda.current_stream(dev)
for blk_type, name, n in ((lb.TPSAP_T_SCH_F, "SCH/F", 4096 * 70), (lb.TPSAP_T_NDB, "NDB half slot", 2 * 4096 * 70),
                          (lb.TPSAP_T_SB1, "SB1", 4096 * 18), (lb.TPSAP_T_SCH_F, "SCH/F, 4 s of signal", 4 * 4096 * 70)):
    p = lb.blk_param(blk_type)
    in_stride, out_stride = p.type345_bits, p.type2_bits
    rows = torch.randint(0, 2, (n, in_stride), dtype=torch.uint8, device=dev, generator=g)
    si = torch.randint(0, 2 ** 31, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    out = torch.zeros((n, out_stride), dtype=torch.uint8, device=dev)
    ok = torch.zeros(n, dtype=torch.int32, device=dev)
    for _ in range(3):
        lb.decode_batch_device(blk_type, rows, n, in_stride, si, out, out_stride, ok, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    reps = 20
    for _ in range(reps):
        lb.decode_batch_device(blk_type, rows, n, in_stride, si, out, out_stride, ok, s)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    steps = p.type2_bits + 4
    print(json.dumps({"kernel": "k_lmac_decode", "block": name, "blocks": n, "ms": round(ms, 4),
                      "Mblocks_per_s": round(n / ms / 1e3, 2), "decoded_Mbit_per_s": round(n * p.type2_bits / ms / 1e3, 1),
                      "G_butterflies_per_s": round(n * steps * 8 / ms / 1e6, 1),
                      "hbm_GBps": round(n * (in_stride + out_stride + 8) / ms / 1e6, 1)}))
