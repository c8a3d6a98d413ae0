"""Times the whole device-resident receive chain on one MI355X, stage by stage, with HIP events (torch.cuda.Event on the
stream every stage is enqueued on): IQ [4096 channels x 36000 samples = 1 s of signal per channel]
  -> tetra_demod_process_device -> tetra_bsync_process_device -> tetra_burst_demux_device x4 -> tetra_lmac_decode_batch_device x4
(SB1, SB2, SCH/F and BBK).  64 distinct synthetic downlinks (training sequences in place, random payload) tiled to 4096
channels; the chain is run over three consecutive seconds of signal with state carried and the third -- demodulator
converged, synchroniser LOCKED on every channel -- is the one timed.
`python profiles/measure_pipeline.py packed` (round 5) hands the frames on PACKED (16 words per frame instead of 512 bytes:
tetra_bsync_process_packed_device -> tetra_burst_demux_packed_device); without the argument the byte-per-bit frames of round 4."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetra_amd  # noqa: E402

pkg = tetra_amd.pkg
lb, bb = pkg.lmac_binding, pkg.bsync_binding
dev = torch.device("cuda", 0)
C, N, SEC, DISTINCT = 4096, 36000, 3, 64
PACKED = "packed" in sys.argv[1:]
stride = pkg.binding.bits_stride(N)
n_slots = SEC * N // 510 + 2
iq_all = np.stack([pkg.synth.gen_channel(SEC * N, 4000 + c, bits=pkg.synth.gen_slot_bits(n_slots, c))[0] for c in range(DISTINCT)])
s = torch.cuda.current_stream(dev)
d = pkg.Demodulator(C, N)
bs = bb.BurstSync(C, stride)
F = bs.max_frames
d_bits = torch.zeros((C, stride), dtype=torch.uint8, device=dev)
d_nbits = torch.zeros(C, dtype=torch.int32, device=dev)
d_frames = torch.zeros((C, F, 16), dtype=torch.int32, device=dev) if PACKED else torch.zeros((C, F, 512), dtype=torch.uint8, device=dev)
d_ft = torch.zeros((C, F), dtype=torch.int32, device=dev)
d_fb = torch.zeros((C, F), dtype=torch.int32, device=dev)
d_nf = torch.zeros(C, dtype=torch.int32, device=dev)
d_scr = torch.full((C * F,), 0x41800117, dtype=torch.int32, device=dev)
kinds = (("SB1", lb.TPSAP_T_SB1, 1, 120, 80), ("SB2", lb.TPSAP_T_SB2, 2, 216, 144), ("SCH/F", lb.TPSAP_T_SCH_F, 0, 432, 288),
         ("BBK", lb.TPSAP_T_BBK, 0, 32, 32))
bufs = {k[0]: (torch.zeros((C * F, k[3]), dtype=torch.uint8, device=dev), torch.zeros(C * F, dtype=torch.int32, device=dev),
               torch.zeros((C * F, k[4]), dtype=torch.uint8, device=dev), torch.zeros(C * F, dtype=torch.int32, device=dev)) for k in kinds}


def ev():
    return torch.cuda.Event(enable_timing=True)


for sec in range(SEC):
    d_iq = torch.from_numpy(np.tile(iq_all[:, sec * N:(sec + 1) * N], (C // DISTINCT, 1))).to(dev)
    torch.cuda.synchronize()
    marks = [ev()]
    marks[0].record(s)
    d.process_device(d_iq, N, d_bits, stride, d_nbits, stream=s)
    marks.append(ev()); marks[-1].record(s)
    (bs.process_packed_device if PACKED else bs.process_device)(d_bits, stride, d_nbits, d_frames, d_ft, d_fb, d_nf, s)
    marks.append(ev()); marks[-1].record(s)
    for name, tpsap, blk, rs, os_ in kinds:
        rows, valid, t2, ok = bufs[name]
        bb.demux_device(d_frames, d_ft, C * F, tpsap, blk, rows, rs, valid, s, packed=PACKED)
    marks.append(ev()); marks[-1].record(s)
    for name, tpsap, blk, rs, os_ in kinds:
        rows, valid, t2, ok = bufs[name]
        lb.decode_batch_device(tpsap, rows, C * F, rs, d_scr, t2, os_, ok, s)
    marks.append(ev()); marks[-1].record(s)
    torch.cuda.synchronize()
    t = [marks[i].elapsed_time(marks[i + 1]) for i in range(4)]
    locked = sum(1 for st in bs.states() if st[0] == bb.RX_S_LOCKED)
    nf = d_nf.cpu().numpy()
    ft = d_ft.cpu().numpy()
    # Rooflines: algorithmic HBM bytes of every stage against 8 TB/s, and -- the decoder's add-compare-select
    # recursion is integer VALU work -- its vector-instruction rate against the chip's issue peak (256 CUs x 4 SIMDs x one wave
    # instruction per 4 clocks at 2.4 GHz = 614 G wave-instructions/s; 68 instructions per trellis step per 64-block wave,
    # DESIGN.md 8.3).  Bytes: demodulator 9 B per sample; synchroniser = the bit rows it scans + the 512-byte frames it writes;
    # a demux launch = frame types + the frames that carry the kind (read) + every row slot (written); a decode launch = every
    # row slot read + type-1 bits, CRC flag and scrambling code per slot.
    n_bits_total = int(d_nbits.sum().item())
    frames_total = int(nf.sum())
    fbytes = 64.0 if PACKED else 512.0
    by = {"demod": 9.0 * C * N, "burst_sync": n_bits_total + fbytes * frames_total + 8.0 * C * F}
    by["demux_x4"] = sum(4.0 * C * F + (float(rs) / 8.0 if PACKED else float(rs)) * int((bufs[name][1] != 0).sum().item()) + float(rs) * C * F
                         for name, _, _, rs, _ in kinds)
    by["lmac_x4"] = sum((float(rs) + os_ + 8.0) * C * F for _, _, _, rs, os_ in kinds)
    steps = {"SB1": 80 + 4, "SB2": 144 + 4, "SCH/F": 288 + 4}          # trellis steps per block incl. the flush (BBK: no Viterbi)
    wave_instr = sum(68.0 * steps[k[0]] * (C * F / 64.0) for k in kinds if k[0] in steps)
    roof = {st_: {"bytes": round(b), "GBps": round(b / (ms * 1e-3) / 1e9, 1), "frac_hbm_8TBps": round(b / (ms * 1e-3) / 8e12, 4)}
            for st_, b, ms in (("demod", by["demod"], t[0]), ("burst_sync", by["burst_sync"], t[1]), ("demux_x4", by["demux_x4"], t[2]),
                               ("lmac_x4", by["lmac_x4"], t[3]))}
    roof["lmac_x4"]["valu_wave_instr_per_s"] = round(wave_instr / (t[3] * 1e-3) / 1e9, 1)
    roof["lmac_x4"]["frac_valu_issue_614G"] = round(wave_instr / (t[3] * 1e-3) / 614.4e9, 4)
    roof["lmac_x4"]["bound"] = "valu-issue (integer add-compare-select)"
    print(json.dumps({"second": sec, "frames_packed": PACKED, "channels": C, "samples_per_channel": N, "ms_demod": round(t[0], 3), "ms_burst_sync": round(t[1], 3),
                      "ms_demux_x4": round(t[2], 3), "ms_lmac_x4": round(t[3], 3), "ms_total": round(sum(t), 3),
                      "x_real_time": round(1000.0 / sum(t), 1), "channels_locked": locked, "frames": int(nf.sum()),
                      "bursts_with_callback": int((ft >= 0).sum()), "frame_slots_decoded_per_kind": C * F, "roofline": roof}))
