"""Soak of the receive chain: 4096 channels, the bench's coded-downlink blocks streamed round and round for `calls` calls on two
streams, fetching one kind per call like a consumer; at the end every block of the last call is CRC-good, all channels are locked,
and the device's free memory is where it was after the first rounds (no growth in the library's pools, nothing leaked per call).
Usage: python profiles/soak_rx_gpu.py [calls] [out.json]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tetra_amd  # noqa: E402

pkg = tetra_amd.pkg
R, synth = pkg.rx_binding, pkg.synth
dev = torch.device("cuda", 0)
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
C, N, SEC, DISTINCT = 4096, 36000, 4, 32
n_slots = SEC * N // 510 + 2
down = [synth.gen_downlink(n_slots, 5100 + c, cell=(300 + c, 2000 + c, c % 64))[0] for c in range(DISTINCT)]
iq = np.stack([synth.gen_channel(SEC * N, 5200 + c, bits=down[c][: 2 * (synth.needed_bits(SEC * N) // 2 + 1)], esn0_db=25.0)[0] for c in range(DISTINCT)])
d_iq = [torch.from_numpy(np.tile(iq[:, k * N:(k + 1) * N], (C // DISTINCT, 1))).to(dev) for k in range(SEC)]
s = torch.cuda.current_stream(dev)
rx = pkg.RxChain(C, N)
free = []
t0 = time.time()
rows = 0
for k in range(calls):
    rx.process_device(d_iq[k % SEC], N, s)
    if k >= 1:
        rows += rx.count(k % R.N_KINDS, which=1)
    if k in (3 * SEC, calls - 1):
        rx.wait()
        free.append(torch.cuda.mem_get_info(dev)[0])
rx.wait()
el = time.time() - t0
good = total = 0
for kind in range(R.N_KINDS):
    blocks, _ = rx.fetch(kind)
    good += int((blocks["crc_ok"] != 0).sum())
    total += len(blocks)
locked = sum(1 for st in rx.sync_states() if st[0] == 2)
res = dict(calls=calls, seconds_of_signal_per_channel=calls, wall_s=round(el, 2), ms_per_call=round(el * 1e3 / calls, 3), rows_counted=rows,
           last_call_blocks=total, last_call_crc_good=good, channels_locked=locked, free_bytes_after_warmup=free[0], free_bytes_at_end=free[-1],
           free_bytes_change=free[-1] - free[0])
rx.close()
print(json.dumps(res))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"))
sys.exit(0 if (good == total and total > 500000 and locked == C and abs(free[-1] - free[0]) < (64 << 20)) else 1)
