"""Same chain as measure_pipeline.py, but software-pipelined over two HIP streams: the demodulator of second k+1 runs on
stream A while burst synchroniser -> demultiplexer -> decoder of second k run on stream B (double-buffered bit rows, events
for the hand-over).  The demodulator leaves ~23 % of the VALU issue slots and half the LDS of every CU free (one 6-wave
workgroup per CU), so the decoder-side kernels can co-reside.  Reports steady-state ms per second of 4096 channels."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetra_amd  # noqa: E402

pkg = tetra_amd.pkg
lb, bb = pkg.lmac_binding, pkg.bsync_binding
dev = torch.device("cuda", 0)
C, N, SEC, DISTINCT = 4096, 36000, 6, 64
stride = pkg.binding.bits_stride(N)
n_slots = SEC * N // 510 + 2
iq_all = np.stack([pkg.synth.gen_channel(SEC * N, 4000 + c, bits=pkg.synth.gen_slot_bits(n_slots, c))[0] for c in range(DISTINCT)])
d_iq = [torch.from_numpy(np.tile(iq_all[:, k * N:(k + 1) * N], (C // DISTINCT, 1))).to(dev) for k in range(SEC)]
PRIO = "prio" in sys.argv[1:]       # experiment: the demodulator's stream at high priority, the decoder side's at low
sA, sB = (torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=0)) if PRIO else (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
d = pkg.Demodulator(C, N)
bs = bb.BurstSync(C, stride)
F = bs.max_frames
d_bits = [torch.zeros((C, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
d_nbits = [torch.zeros(C, dtype=torch.int32, device=dev) for _ in range(2)]
d_frames = torch.zeros((C, F, 16), dtype=torch.int32, device=dev) if "packed" in sys.argv[1:] else torch.zeros((C, F, 512), dtype=torch.uint8, device=dev)
d_ft = torch.zeros((C, F), dtype=torch.int32, device=dev)
d_fb = torch.zeros((C, F), dtype=torch.int32, device=dev)
d_nf = torch.zeros(C, dtype=torch.int32, device=dev)
d_scr = torch.full((C * F,), 0x41800117, dtype=torch.int32, device=dev)
kinds = (("SB1", lb.TPSAP_T_SB1, 1, 120, 80), ("SB2", lb.TPSAP_T_SB2, 2, 216, 144), ("SCH/F", lb.TPSAP_T_SCH_F, 0, 432, 288),
         ("BBK", lb.TPSAP_T_BBK, 0, 32, 32))
bufs = {k[0]: (torch.zeros((C * F, k[3]), dtype=torch.uint8, device=dev), torch.zeros(C * F, dtype=torch.int32, device=dev),
               torch.zeros((C * F, k[4]), dtype=torch.uint8, device=dev), torch.zeros(C * F, dtype=torch.int32, device=dev)) for k in kinds}


cbufs = {k[0]: (torch.zeros((C * F,), dtype=torch.int32, device=dev), torch.zeros((1,), dtype=torch.int32, device=dev)) for k in kinds}
PACKED = "packed" in sys.argv[1:]      # round 5: frames handed on as 16 words instead of 512 bytes
COMPACT = True      # set per run: rows only for the frames that carry the kind (tetra_burst_demux_compact_device + counted decoder)


def chain(b, stream):
    (bs.process_packed_device if PACKED else bs.process_device)(d_bits[b], stride, d_nbits[b], d_frames, d_ft, d_fb, d_nf, stream)
    for name, tpsap, blk, rs, os_ in kinds:
        rows, valid, t2, ok = bufs[name]
        if COMPACT:
            idx, cnt = cbufs[name]
            bb.demux_compact_device(d_frames, d_ft, C * F, tpsap, blk, rows, rs, idx, cnt, stream, packed=PACKED)
            lb.decode_counted_device(tpsap, rows, C * F, cnt, rs, d_scr, idx, t2, os_, ok, stream)
        else:
            bb.demux_device(d_frames, d_ft, C * F, tpsap, blk, rows, rs, valid, stream, packed=PACKED)
            lb.decode_batch_device(tpsap, rows, C * F, rs, d_scr, t2, os_, ok, stream)


def run(overlap):
    d.reset()
    bs.reset()
    torch.cuda.synchronize()
    ev_demod = [torch.cuda.Event() for _ in range(SEC)]
    ev_chain = [torch.cuda.Event() for _ in range(SEC)]
    t0 = time.perf_counter()
    for k in range(SEC + 1):
        if k < SEC:
            if k >= 2:
                sA.wait_event(ev_chain[k - 2])          # bits[k % 2] is free again
            d.process_device(d_iq[k], N, d_bits[k % 2], stride, d_nbits[k % 2], stream=sA)
            ev_demod[k].record(sA)
        if k >= 1:
            s = sB if overlap else sA
            s.wait_event(ev_demod[k - 1])
            chain((k - 1) % 2, s)
            ev_chain[k - 1].record(s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / SEC


run(True)                                                 # warm-up (allocator pools, clocks)
run(True)
res = {"channels": C, "samples_per_channel": N, "seconds": SEC}
for name, ov, comp in (("serial_one_stream", False, False), ("overlapped_two_streams", True, False),
                       ("serial_one_stream_compact", False, True), ("overlapped_two_streams_compact", True, True),
                       ("serial_one_stream_again", False, False), ("overlapped_two_streams_again", True, False),
                       ("serial_one_stream_compact_again", False, True), ("overlapped_two_streams_compact_again", True, True)):
    COMPACT = comp
    ms = run(ov)
    res[name + "_ms_per_second"] = round(ms, 3)
    res[name + "_x_real_time"] = round(1000.0 / ms, 1)
res["rows_decoded_per_kind_compact"] = {k[0]: int(cbufs[k[0]][1].cpu().numpy()[0]) for k in kinds}
res["rows_decoded_per_kind_slot_layout"] = C * F
res["channels_locked"] = sum(1 for st in bs.states() if st[0] == bb.RX_S_LOCKED)
print(json.dumps(res))
