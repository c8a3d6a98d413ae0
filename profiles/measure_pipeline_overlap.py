"""The receive chain through its one handle (include/tetra_rx.h), steady state: `python profiles/measure_pipeline_overlap.py` prints
bench.py's `chain` object (coded downlinks, two streams / one stream, stage times, known-answer counters).  `random` = round 5's
workload instead (training sequences in place, RANDOM payload, SB1 + SB2 + SCH/F + BBK only: comparable with profiles/r05/r05_l_*)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import tetra_amd  # noqa: E402

pkg = tetra_amd.pkg
dev = torch.device("cuda", 0)
if "random" not in sys.argv[1:]:
    print(json.dumps(bench.receive_chain(None, torch, pkg, dev, 0)))
    sys.exit(0)
R = pkg.rx_binding
C, N, SEC, DISTINCT = 4096, 36000, 6, 64
n_slots = SEC * N // 510 + 2
iq_all = np.stack([pkg.synth.gen_channel(SEC * N, 4000 + c, bits=pkg.synth.gen_slot_bits(n_slots, c))[0] for c in range(DISTINCT)])
d_iq = [torch.from_numpy(np.tile(iq_all[:, k * N:(k + 1) * N], (C // DISTINCT, 1))).to(dev) for k in range(SEC)]
s = torch.cuda.current_stream(dev)
res = {"channels": C, "samples_per_channel": N, "seconds": SEC, "payload": "random (round 5's workload)"}
for name, flags in (("overlapped_two_streams", 0), ("serial_one_stream", R.FLAG_ONE_STREAM)):
    rx = pkg.RxChain(C, N, flags=flags, kinds=(1 << R.KIND_SB1) | (1 << R.KIND_SB2) | (1 << R.KIND_SCH_F) | (1 << R.KIND_BBK))
    for rep in range(3):
        rx.wait()
        t0 = time.perf_counter()
        for k in range(SEC):
            rx.process_device(d_iq[k], N, s)
        rx.wait()
        ms = (time.perf_counter() - t0) * 1e3 / SEC
    res[name + "_ms_per_second"] = round(ms, 3)
    res[name + "_stage_ms"] = [round(v, 4) for v in rx.stage_ms()]
    res[name + "_rows"] = {k: rx.count(k) for k in (R.KIND_SB1, R.KIND_SB2, R.KIND_SCH_F, R.KIND_BBK)}
    rx.close()
print(json.dumps(res))
