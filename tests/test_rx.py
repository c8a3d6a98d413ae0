"""The receive chain behind one handle (include/tetra_rx.h): IQ in, decoded type-1 blocks + CRC + TDMA time + cell state out.

Reference anchors: tetra_burst_sync_in (phy/tetra_burst_sync.c:54-155) -> tetra_burst_rx_cb (phy/tetra_burst.c:343-393) ->
tp_sap_udata_ind (lower_mac/tetra_lower_mac.c:148-275).  The transmit side of these tests is the reference's own encoder
primitives and burst builders (oracle/_ref) or synth.gen_downlink, which tests/test_synth_tx.py holds bit for bit against them."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rx_header_symbols_all_exported(pkg):
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "tetra_rx.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(tetra_rx_[a-z0-9_]+)\s*\(", src)))
    L = pkg.load_library()
    assert set(names) == set(pkg.rx_binding.RX_EXPORTS) and len(names) <= 16
    for n in names:
        assert hasattr(L, n), n


def test_rx_struct_layouts_match_header(pkg):
    prog = ('#include <stdio.h>\n#include "tetra_rx.h"\nint main(){printf("%zu %zu %zu %d %d\\n", sizeof(tetra_rx_config_t), '
            'sizeof(tetra_rx_block_t), sizeof(tetra_lmac_cell_state_t), TETRA_RX_KIND_SCH_F, TETRA_RX_N_KINDS);return 0;}\n')
    with tempfile.TemporaryDirectory() as td:
        cfile = os.path.join(td, "s.c")
        open(cfile, "w").write(prog)
        exe = os.path.join(td, "s")
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), cfile, "-o", exe], check=True)
        a, b, c, k, n = map(int, subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split())
    R = pkg.rx_binding
    assert a == C.sizeof(R.RxConfig) and b == C.sizeof(R.RxBlock) == R.BLOCK_DTYPE.itemsize and c == C.sizeof(R.CellState)
    assert k == R.KIND_SCH_F and n == R.N_KINDS
    assert [R.type1_bits(k) for k in range(6)] == [60, 30, 124, 124, 124, 268] and R.type1_bits(6) < 0


def test_rx_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.TetraDemodError):
        pkg.RxChain(4, 1000)


def _uint_bits(v, n):
    return [(v >> (n - 1 - i)) & 1 for i in range(n)]


@pytest.mark.gpu
def test_gpu_rx_iq_to_type1_blocks_through_one_handle(pkg, ref, synth):
    """tests/test_burst_sync.py::test_gpu_iq_to_type1_blocks_all_on_device re-expressed through the handle, identical assertions:
    transmit side = the reference's encoder primitives and burst builders; every channel a different cell (its SYNC PDUs carry its
    own MCC / MNC / colour code, its SCH/F blocks are scrambled with the code the reference derives from them); nothing but IQ goes
    in, ONE call; the SYNC PDUs and the SCH/F blocks the reference would hand to its upper MAC come back with good CRCs, and the
    handle's cell state holds the code the reference computes."""
    if not ref.lmac_available():
        pytest.skip("oracle/_ref/libtetra_lmac_ref.so not available")
    R = pkg.rx_binding
    rng = np.random.default_rng(21)
    Cn, nslots = 8, 44
    cells = [(int(rng.integers(0, 1024)), int(rng.integers(0, 16384)), int(rng.integers(0, 64))) for _ in range(Cn)]
    codes = [ref.scramb_get_init(*cell) for cell in cells]
    sent_sb1, sent_schf, tx = [set() for _ in range(Cn)], [set() for _ in range(Cn)], []
    for c in range(Cn):
        mcc, mnc, cc = cells[c]
        slots = []
        for s in range(nslots):
            bbk = rng.integers(0, 2, 30)
            if s % 4 == 0:
                t1 = rng.integers(0, 2, 60).astype(np.uint8)
                t1[4:10], t1[31:41], t1[41:55] = _uint_bits(cc, 6), _uint_bits(mcc, 10), _uint_bits(mnc, 14)
                sent_sb1[c].add(t1.tobytes())
                slots.append(ref.build_sync_burst(ref.lmac_encode(ref.TPSAP_T_SB1, t1, 3), bbk, rng.integers(0, 2, 216)))
            else:
                t1 = rng.integers(0, 2, 268).astype(np.uint8)
                sent_schf[c].add(t1.tobytes())
                t5 = ref.lmac_encode(ref.TPSAP_T_SCH_F, t1, codes[c])
                slots.append(ref.build_norm_burst(t5[:216], bbk, t5[216:], 0))
        tx.append(np.concatenate(slots))
    N = nslots * 510 - 100
    iq = np.stack([synth.gen_channel(N, 500 + c, bits=tx[c])[0] for c in range(Cn)])
    rx = pkg.RxChain(Cn, N)
    rx.process(iq)
    rx.wait()
    states, cell = rx.sync_states(), rx.cells()
    out = {"sb1": rx.fetch(R.KIND_SB1), "schf": rx.fetch(R.KIND_SCH_F)}
    for c in range(Cn):
        assert states[c][0] == pkg.bsync_binding.RX_S_LOCKED
        assert cell[c].scramb_init == codes[c] and (cell[c].mcc, cell[c].mnc, cell[c].colour_code) == cells[c]
        for name, sent in (("sb1", sent_sb1), ("schf", sent_schf)):
            blocks, t1 = out[name]
            mine = blocks["channel"] == c
            good = mine & (blocks["crc_ok"] != 0)
            assert good.sum() >= (4 if name == "sb1" else 12), (c, name, int(good.sum()))
            assert all(t1[j].tobytes() in sent[c] for j in np.nonzero(good)[0])
            if name == "sb1":
                assert mine.sum() - good.sum() <= 1                # at most the first frame after lock may still be settling
        # rows come in (channel, frame) order, bit numbers ascending within a channel, 510 apart while locked
        blocks = out["schf"][0]
        bn = blocks["bitnum"][blocks["channel"] == c].astype(np.int64)
        assert (np.diff(bn) > 0).all() and (np.diff(bn) % 510 == 0).all()
    assert (np.diff(out["schf"][0]["channel"]) >= 0).all()
    rx.close()


def _downlink_batch(synth, Cn, nslots, N, seed):
    cells = [(100 + 7 * c, 1000 + 13 * c, (5 + 3 * c) % 64) for c in range(Cn)]
    tx = [synth.gen_downlink(nslots, seed + c, cell=cells[c]) for c in range(Cn)]
    iq = np.stack([synth.gen_channel(N, seed + 100 + c, bits=tx[c][0])[0] for c in range(Cn)])
    return cells, tx, iq


def _collect(rx, R, which=0):
    got = {}
    for k in range(R.N_KINDS):
        blocks, t1 = rx.fetch(k, which)
        got[k] = [(int(b["channel"]), int(b["bitnum"]), int(b["crc_ok"]), int(b["tdma_time_rx"]), int(b["tdma_time"]), t1[j].tobytes())
                  for j, b in enumerate(blocks)]
    return got


@pytest.mark.gpu
def test_gpu_rx_all_kinds_cell_and_clock(pkg, synth, ref):
    """A coded downlink with every block kind (synth.gen_downlink: SYNC bursts with SB1 + AACH + SB2, two-channel normal bursts with
    NDB blocks 1 + 2, one-channel normal bursts with SCH/F; a different cell per channel, TDMA time in the SYNC PDUs) through the
    handle: after lock every block comes back with a good CRC and the type-1 bits that were sent IN THE SLOT its TDMA time names
    -- i.e. the tracker's clock (tetra_burst_sync.c:113 + tetra_lower_mac.c:246-275) labels every frame with the time the
    transmitter gave it --, the AACH bits descrambled with the cell's code, and the cell state reads the cell."""
    R = pkg.rx_binding
    Cn, nslots = 6, 80
    N = nslots * 510 - 100
    cells, tx, iq = _downlink_batch(synth, Cn, nslots, N, 3000)
    rx = pkg.RxChain(Cn, N)
    rx.process(iq)
    rx.wait()
    got = _collect(rx, R)
    cell = rx.cells()
    names = {R.KIND_SB1: "sb1", R.KIND_BBK: "bbk", R.KIND_SB2: "sb2", R.KIND_NDB1: "ndb1", R.KIND_NDB2: "ndb2", R.KIND_SCH_F: "schf"}
    for c in range(Cn):
        assert (cell[c].mcc, cell[c].mnc, cell[c].colour_code) == cells[c]
        assert cell[c].scramb_init == synth.tx_scramb_code(*cells[c])
        by_time = {}
        for s in range(nslots):
            tn, fn, mn = synth.tdma_time_of_slot(s)
            by_time[tn | fn << 8 | mn << 16] = s
        exact = 0
        for k, name in names.items():
            sent = {s: v.tobytes() for s, v in tx[c][1][name]}
            rows = [r for r in got[k] if r[0] == c]
            # blocks after the first good SYNC PDU: handled under the cell's code and the transmitter's clock
            first = min(r[1] for r in got[R.KIND_SB1] if r[0] == c and r[2])
            rows = [r for r in rows if r[1] > first]
            assert len(rows) >= {"sb1": 8, "sb2": 8, "ndb1": 8, "ndb2": 8, "schf": 20, "bbk": 50}[name], (c, name, len(rows))
            for ch, bitnum, ok, t_rx, t, bits in rows:
                # The demodulator's loops take a dozen slots to settle at 25 dB.  Until then a frame may be lost (the receiver falls
                # back to UNLOCKED and the clock, like the reference's, is only right again after the next good SYNC PDU) and a block
                # may fail its CRC: there, a good CRC must still mean one of the blocks that were sent; from slot 28 on everything
                # is exact: CRC good, the bits of the very slot the block's TDMA time names.
                if bitnum < 28 * 510:
                    if ok and name != "bbk":      # (the AACH is handed on uncoded, tetra_lower_mac.c:231-236: no CRC to tell)
                        assert bits in sent.values(), (c, name, bitnum)
                    continue
                assert ok == 1 and t in by_time, (c, name, bitnum, hex(t))
                assert sent[by_time[t]] == bits, (c, name, by_time[t])
                exact += 1
        assert exact >= 5 * (nslots - 32) // 2, (c, exact)          # 3 + 2 + 3 + 2 blocks per four slots from slot 28 on
    rx.close()


@pytest.mark.gpu
def test_gpu_rx_equals_the_separate_stage_entry_points(pkg, synth):
    """The handle is the composition of the pinned stages, nothing else: the same IQ through tetra_demod_process_device ->
    tetra_bsync_process_device -> tetra_burst_demux_device -> tetra_lmac_decode_batch_device (slot layout, byte frames) with
    tetra_lmac_track_sync_device in between gives, row for row, the handle's blocks: type-1 bits, crc, bit numbers, both times."""
    import torch
    R, lb, bb_ = pkg.rx_binding, pkg.lmac_binding, pkg.bsync_binding
    Cn, nslots = 5, 60
    N = nslots * 510 - 300
    cells, tx, iq = _downlink_batch(synth, Cn, nslots, N, 4000)
    rx = pkg.RxChain(Cn, N)
    rx.process(iq)
    rx.wait()
    got = _collect(rx, R)
    rx.close()
    dev = torch.device("cuda", 0)
    d = pkg.Demodulator(Cn, N)
    stride = d.bits_stride(N)
    bs = bb_.BurstSync(Cn, stride)
    F = bs.max_frames
    z = lambda shape, dt=torch.int32: torch.zeros(shape, dtype=dt, device=dev)
    d_bits, d_nbits = z((Cn, stride), torch.uint8), z(Cn)
    d_frames, d_ft, d_fb, d_nf = z((Cn, F, 512), torch.uint8), z((Cn, F)), z((Cn, F)), z(Cn)
    d.process_device(torch.from_numpy(iq).to(dev), N, d_bits, stride, d_nbits)
    bs.process_device(d_bits, stride, d_nbits, d_frames, d_ft, d_fb, d_nf)
    rows, valid1, sb1, ok1 = z((Cn * F, 432), torch.uint8), z(Cn * F), z((Cn * F, 80), torch.uint8), z(Cn * F)
    bb_.demux_device(d_frames, d_ft, Cn * F, lb.TPSAP_T_SB1, 1, rows, 432, valid1)
    lb.decode_batch_device(lb.TPSAP_T_SB1, rows, Cn * F, 432, None, sb1, 80, ok1)
    d_cell, d_scr, d_trx, d_t = z((Cn, 10)), z(Cn * F), z(Cn * F), z(Cn * F)
    lb.track_sync_device(sb1, 80, ok1, valid1, d_nf, Cn, F, d_cell, d_scr, d_trx, d_t)
    fb, trx, tt = d_fb.cpu().numpy().view(np.uint32).reshape(-1), d_trx.cpu().numpy().view(np.uint32), d_t.cpu().numpy().view(np.uint32)
    spec = {R.KIND_SB1: (lb.TPSAP_T_SB1, 1, 80), R.KIND_BBK: (lb.TPSAP_T_BBK, 0, 32), R.KIND_SB2: (lb.TPSAP_T_SB2, 2, 144),
            R.KIND_NDB1: (lb.TPSAP_T_NDB, 1, 144), R.KIND_NDB2: (lb.TPSAP_T_NDB, 2, 144), R.KIND_SCH_F: (lb.TPSAP_T_SCH_F, 0, 288)}
    total = 0
    for k, (tpsap, blk, os_) in spec.items():
        valid, t2, ok = z(Cn * F), z((Cn * F, os_), torch.uint8), z(Cn * F)
        bb_.demux_device(d_frames, d_ft, Cn * F, tpsap, blk, rows, 432, valid)
        lb.decode_batch_device(tpsap, rows, Cn * F, 432, None if k == R.KIND_SB1 else d_scr, t2, os_, ok)
        torch.cuda.synchronize()
        v, t2h, okh = valid.cpu().numpy().astype(bool), t2.cpu().numpy(), ok.cpu().numpy()
        n1 = R.type1_bits(k)
        want = [(f // F, int(fb[f]), int(okh[f]), int(trx[f]), int(tt[f]), t2h[f, :n1].tobytes()) for f in np.nonzero(v)[0]]
        assert want == got[k], k
        total += len(want)
    assert total > Cn * nslots
    d.close()
    bs.close()


@pytest.mark.gpu
def test_gpu_rx_streaming_calls_overlap_and_equal_one_call(pkg, synth):
    """The stream cut into ragged calls (two in flight: the demodulator of call k+1 beside the tail of call k, results double
    buffered and fetched one call late) delivers exactly the blocks of ONE call over the whole stream, and so does the one-stream
    mode (TETRA_RX_FLAG_ONE_STREAM); kinds can be masked."""
    import torch
    R = pkg.rx_binding
    Cn, nslots = 4, 90
    N = nslots * 510 - 100
    cells, tx, iq = _downlink_batch(synth, Cn, nslots, N, 5000)
    one = pkg.RxChain(Cn, N)
    one.process(iq)
    one.wait()
    want = _collect(one, R)
    want_cell = [tuple(getattr(c, f) for f, _ in R.CellState._fields_) for c in one.cells()]
    one.close()
    dev = torch.device("cuda", 0)
    d_iq = torch.from_numpy(iq).to(dev)
    cuts = [0, 9000, 9001, 20000, 20180, 33000, N]
    for flags in (0, R.FLAG_ONE_STREAM):
        rx = pkg.RxChain(Cn, 16000, flags=flags)
        s = torch.cuda.Stream(dev)
        got = {k: [] for k in range(R.N_KINDS)}
        for i, (a, b) in enumerate(zip(cuts, cuts[1:])):
            chunk = d_iq[:, a:b].contiguous()
            s.wait_stream(torch.cuda.current_stream(dev))
            rx.process_device(chunk, b - a, s)
            if i >= 1:                                   # the previous call's results while this one runs
                prev = _collect(rx, R, which=1)
                for k in got:
                    got[k] += prev[k]
            chunk.record_stream(s)
        last = _collect(rx, R, which=0)
        for k in got:
            got[k] += last[k]
        rx.wait()
        for k in got:
            assert sorted(got[k]) == sorted(want[k]), (flags, k, len(got[k]), len(want[k]))
        assert [tuple(getattr(c, f) for f, _ in R.CellState._fields_) for c in rx.cells()] == want_cell
        ms = rx.stage_ms()
        assert len(ms) == 4 and all(m >= 0 for m in ms)
        rx.close()
    rx = pkg.RxChain(Cn, N, kinds=1 << R.KIND_SCH_F)
    rx.process(iq)
    assert _collect_kind(rx, R, R.KIND_SCH_F) == want[R.KIND_SCH_F] and _collect_kind(rx, R, R.KIND_SB1) == want[R.KIND_SB1]
    with pytest.raises(pkg.TetraDemodError) as e:
        rx.fetch(R.KIND_SB2)
    assert e.value.status == -2          # TETRA_ERR_UNSUPPORTED
    rx.close()


@pytest.mark.gpu
def test_gpu_rx_time_major_frames_equal_channel_major(pkg, synth):
    """The chain behind the channeliser / resampler: the same signals handed over as time-major frames [sample][channel]
    (TETRA_LAYOUT_TIME_MAJOR, what tetra_chan / tetra_resamp emit) give exactly the blocks, labels and cell states of the
    channel-major handle."""
    import torch
    R = pkg.rx_binding
    Cn, nslots = 6, 40
    N = nslots * 510
    cells, tx, iq = _downlink_batch(synth, Cn, nslots, N, 7100)
    a = pkg.RxChain(Cn, N)
    a.process(iq)
    a.wait()
    want = _collect(a, R)
    want_cell = [tuple(getattr(c, f) for f, _ in R.CellState._fields_) for c in a.cells()]
    a.close()
    dev = torch.device("cuda", 0)
    d_tm = torch.from_numpy(np.ascontiguousarray(iq.T)).to(dev)            # [N][C]
    b = pkg.RxChain(Cn, N, layout=pkg.binding.LAYOUT_TIME_MAJOR)
    b.process_device(d_tm, N, torch.cuda.current_stream(dev))
    b.wait()
    got = _collect(b, R)
    assert sum(len(v) for v in want.values()) > 100
    for k in want:
        assert got[k] == want[k], k
    assert [tuple(getattr(c, f) for f, _ in R.CellState._fields_) for c in b.cells()] == want_cell
    b.close()


def _collect_kind(rx, R, k):
    blocks, t1 = rx.fetch(k)
    return [(int(b["channel"]), int(b["bitnum"]), int(b["crc_ok"]), int(b["tdma_time_rx"]), int(b["tdma_time"]), t1[j].tobytes())
            for j, b in enumerate(blocks)]


@pytest.mark.gpu
def test_gpu_rx_statuses_reset_and_device_views(pkg, synth):
    import torch
    R = pkg.rx_binding
    L = pkg.load_library()
    Cn, nslots = 3, 40
    N = nslots * 510
    cells, tx, iq = _downlink_batch(synth, Cn, nslots, N, 6000)
    rx = pkg.RxChain(Cn, N)
    assert rx.count(R.KIND_SB1) == 0 and rx.count(R.KIND_SB1, which=1) == 0            # before any call
    assert rx.max_rows == Cn * ((4096 + pkg.binding.bits_stride(N)) // 510 + 2) or rx.max_rows > 0
    with pytest.raises(pkg.TetraDemodError) as e:
        rx.process(np.zeros((Cn, N + 1), np.complex64))
    assert e.value.status == -6          # TETRA_ERR_SIZE
    rx.process(iq)
    n = rx.count(R.KIND_SCH_F)
    assert n > 10
    got = C.c_int(0)
    small = np.zeros(3, R.BLOCK_DTYPE)
    rc = L.tetra_rx_fetch(rx._h, 0, R.KIND_SCH_F, small.ctypes.data_as(C.c_void_p), None, 0, 3, C.byref(got))
    assert rc == -6 and got.value == n                                                     # too small: a status and the needed size
    t1 = np.zeros((n, 100), np.uint8)
    rc = L.tetra_rx_fetch(rx._h, 0, R.KIND_SCH_F, None, t1.ctypes.data_as(C.c_void_p), 100, n, C.byref(got))
    assert rc == -6                                                                        # type1_stride < 268
    assert L.tetra_rx_fetch(rx._h, 2, R.KIND_SCH_F, None, None, 0, 0, C.byref(got)) == -1 and L.tetra_rx_fetch(rx._h, 0, 9, None, None, 0, 0, C.byref(got)) == -1
    # device views: same rows as the host fetch
    blocks, t1 = rx.fetch(R.KIND_SCH_F)
    p_t2, st, p_blk, p_n = rx.rows_device(R.KIND_SCH_F)
    assert st == 288 and p_t2 and p_blk and p_n
    bits_p, bstride, nb_p = rx.bits_device()
    assert bits_p and nb_p and bstride >= N
    # reset: fresh receivers, the same stream gives the same blocks again
    rx.reset()
    assert rx.count(R.KIND_SCH_F) == 0 and all(c.scramb_init == 0 for c in rx.cells())
    rx.process(iq)
    b2, t2 = rx.fetch(R.KIND_SCH_F)
    assert np.array_equal(b2, blocks) and np.array_equal(t2, t1)
    # the demodulator inside takes the reference's setters
    rx.wait()
    assert L.tetra_demod_set_param(C.c_void_p(rx.demod_handle()), 4, C.c_double(0.03)) == 0
    rx.close()
