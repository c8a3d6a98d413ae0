"""The C++ dsp::block mirror (sdrpp-tetra-demodulator_amd/host/): builds against the C ABI with g++;
on a GPU it is driven through streams/worker threads like the reference plugin drives its blocks."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PK = os.path.join(ROOT, "sdrpp-tetra-demodulator_amd")
EXE = os.path.join(ROOT, "tests", "host", "test_block")


def _build(pkg):
    pkg.build.build()
    srcs = [os.path.join(ROOT, "tests", "host", "test_block.cpp")] + [os.path.join(PK, "host", f) for f in
                                                                       ("pi4dqpsk_gpu.cpp", "dqpsk_sym_extr_gpu.cpp", "bit_unpacker_gpu.cpp")]
    deps = srcs + [os.path.join(PK, "host", f) for f in ("pi4dqpsk_gpu.h", "dqpsk_sym_extr_gpu.h", "bit_unpacker_gpu.h", "dsp_compat.h")] + \
        [os.path.join(ROOT, "include", "tetra_demod.h"), os.path.join(ROOT, "tests", "host", "tap_selftest.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        # libamdhip64 only for the test driver's own device buffers ("multibank-device"); the mirror itself needs just the C ABI
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + srcs +
                       ["-L", PK, "-ltetra_demod_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + PK, "-Wl,-rpath,/opt/rocm/lib",
                        "-o", EXE], check=True)
    return EXE


def test_block_mirror_builds_and_links(pkg):
    exe = _build(pkg)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "abi 6" in out and "default taps 65" in out and "decision tap ok" in out and "decision tap alignment ok" in out


@pytest.mark.gpu
def test_block_mirror_streams_like_the_plugin(pkg, oracle, synth, tmp_path):
    """PI4DQPSK (C++ block, worker thread, 180-sample chunks as SDR++ delivers them) -> symbols; slicing the
    symbols the way the plugin's own extractor does gives the oracle's bits."""
    exe = _build(pkg)
    N = 18000
    iq, _, _ = synth.gen_channel(N, 77)
    f_in = tmp_path / "iq.f32"
    iq.view(np.float32).tofile(f_in)
    f_sym, f_bits = tmp_path / "sym.f32", tmp_path / "bits.u8"
    subprocess.run([exe, str(f_in), "180", str(f_sym), str(f_bits)], check=True, timeout=300)
    sym = np.fromfile(f_sym, np.float32).view(np.complex64)
    bits = np.fromfile(f_bits, np.uint8)
    r = oracle.Oracle().process(iq)
    assert np.array_equal(sym.view(np.uint32), r["sym"].view(np.uint32))
    assert np.array_equal(bits, r["bits"])


@pytest.mark.gpu
def test_three_mirrored_blocks_stream_like_the_plugin(pkg, oracle, synth, tmp_path):
    """PI4DQPSK -> DQPSKSymbolExtractor -> BitUnpacker, all three the GPU-backed mirrors
    (/root/reference src/dsp/dqpsk_sym_extr.h:19-46, bit_unpacker.h:16-34; wired as src/main.cpp:84-91), each on its own worker
    thread, 180-sample chunks from a file as SDR++ delivers them (BASELINE config 1's shape).  The sink's bits equal the
    oracle's, the extractor's public standarderr / sync (the GUI's meter, src/main.cpp:211,215) equal the oracle's faithful
    restatement of the statistic within 2e-6 -- and no DSP arithmetic ran on the host (the mirrors only move the kernels'
    decisions).  The blocks' own process() signatures called directly give the same dibits / bits."""
    exe = _build(pkg)
    N = 36000
    iq, _, _ = synth.gen_channel(N, 78)
    f_in = tmp_path / "iq.f32"
    iq.view(np.float32).tofile(f_in)
    f_bits, f_dib, f_sym = tmp_path / "bits.u8", tmp_path / "dib.u8", tmp_path / "sym.f32"
    r = subprocess.run([exe, "chain3", str(f_in), "180", str(f_bits), str(f_dib), str(f_sym)], timeout=300, capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    o = oracle.Oracle()
    want = o.process(iq)
    bits = np.fromfile(f_bits, np.uint8)
    sym = np.fromfile(f_sym, np.float32).view(np.complex64)
    assert np.array_equal(bits, want["bits"]) and np.array_equal(sym.view(np.uint32), want["sym"].view(np.uint32))
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("standarderr")][0].split()
    err, sync, statuses = float(line[1]), int(line[3]), [int(x) for x in line[5:8]]
    assert statuses == [0, 0, 0]
    assert want["sym"].size >= 4096 + 256                                   # the ring is full of real distances by the end
    assert abs(err - float(o.st.standarderr)) < 2e-6 and sync == int(o.st.sync) == 1, (err, float(o.st.standarderr))
    # direct calls of the two blocks' process() on a further 180 samples
    w2 = o.process(iq[:180])
    raw = np.fromfile(f_dib, np.uint8)
    ns = w2["dibits"].size
    assert raw.size == 3 * ns and np.array_equal(raw[:ns], w2["dibits"]) and np.array_equal(raw[ns:], w2["bits"])


@pytest.mark.gpu
def test_decision_tap_realigns_when_a_symbol_buffer_is_lost_or_repeated(pkg, oracle, synth, tmp_path):
    """VERDICT r4 weak 7 / next 6.  The three mirrored blocks on their worker threads as above, but the relay that stands where
    SDR++'s splitter is (src/main.cpp:85-90) LOSES symbol buffer 20 and hands buffer 60 on twice -- what re-binding the splitter
    or disable() / enable() (src/main.cpp:130-167) can do to the blocks behind it.  A positional side channel would hand every
    later symbol another symbol's decision for ever; the self-checking one finds the gap (one resync, exactly the lost buffer's
    symbols skipped), slices the repeated buffer from its own signs (one fallback) and is back on the kernels' decisions with the
    next buffer: the sink's bits for every buffer the extractor was handed equal the oracle's for that buffer (the repeated copy's
    first dibit excepted: a difference against that buffer's own last symbol, in the reference's block too), and the statistic
    marks still arrive at their stream positions (standarderr == the oracle's at the end)."""
    exe = _build(pkg)
    N, chunk = 36000, 180
    iq, _, _ = synth.gen_channel(N, 79)
    f_in = tmp_path / "iq.f32"
    iq.view(np.float32).tofile(f_in)
    f_bits, f_dib, f_sym, f_cnt = tmp_path / "bits.u8", tmp_path / "dib.u8", tmp_path / "sym.f32", tmp_path / "cnt.i32"
    r = subprocess.run([exe, "chain3", str(f_in), str(chunk), str(f_bits), str(f_dib), str(f_sym), "20", "60", str(f_cnt)], timeout=300,
                       capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    o = oracle.Oracle()
    per_chunk = [o.process(iq[p:p + chunk])["bits"] for p in range(0, N, chunk)]
    fwd = np.fromfile(f_cnt, np.int32).reshape(-1, 2)
    assert [int(k) for k in fwd[:, 0]] == [k for k in range(N // chunk) if k != 20 for _ in range(2 if k == 60 else 1)]
    bits = np.fromfile(f_bits, np.uint8)
    pos, seen = 0, set()
    for k, ns in fwd:
        want = per_chunk[int(k)]
        assert 2 * int(ns) == want.size
        got = bits[pos:pos + want.size]
        first = 2 if int(k) in seen else 0          # the repeated copy: its first dibit is sliced against its own last symbol
        assert np.array_equal(got[first:], want[first:]), int(k)
        seen.add(int(k))
        pos += want.size
    assert pos == bits.size
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("extractor resyncs")][0].split()
    resyncs, skipped, fallbacks, u_resyncs, u_fallbacks = int(line[2]), int(line[4]), int(line[6]), int(line[9]), int(line[11])
    assert (resyncs, skipped, fallbacks) == (1, per_chunk[20].size // 2, 1), line
    # the unpacker sits behind the extractor: it sees the extractor's dibits, i.e. the same gap and the same repeated buffer
    assert u_resyncs == 1 and u_fallbacks == 1, line
    st = [ln for ln in r.stdout.splitlines() if ln.startswith("standarderr")][0].split()
    assert abs(float(st[1]) - float(o.st.standarderr)) < 2e-6 and int(st[3]) == int(o.st.sync) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["multibank", "multibank-cs16", "multibank-device"])
def test_multibank_two_shards_two_threads_one_device(pkg, oracle, synth, tmp_path, mode):
    """PI4DQPSKMultiBank with devices {0, 0}: two handles on ONE GPU, each driven by its own host thread through its own
    HIP streams at the same time (the "one handle per thread, no global state" claim of include/tetra_demod.h), over five
    calls with a setter in the middle.  Every channel's bits equal the oracle's; the split follows shard.channel_range.
    Three entry points: float IQ from host memory (process), int16 IQ from host memory (processCS16; the oracle sees the
    dequantised samples), and samples / bit rows resident on the shards' GPUs (processDevice -> tetra_demod_process_resident,
    the node-level form of the path the throughput metric is quoted on)."""
    exe = _build(pkg)
    Cn, n, calls = 37, 20000, 5
    iq, _, _ = synth.gen_batch(Cn, n, base_seed=4100)
    iq = (iq * np.float32(0.5)).astype(np.complex64)          # inside the int16 range
    f_in = tmp_path / "iq.f32"
    np.ascontiguousarray(iq).view(np.float32).tofile(f_in)
    f_bits, f_nb = tmp_path / "bits.u8", tmp_path / "nb.i32"
    r = subprocess.run([exe, mode, str(f_in), str(Cn), str(n), str(calls), str(f_bits), str(f_nb), "0", "0"],
                       check=True, timeout=300, capture_output=True, text=True)
    lo0, hi0 = pkg.shard.channel_range(Cn, 2, 0)
    lo1, hi1 = pkg.shard.channel_range(Cn, 2, 1)
    assert "shard 0: channels [%d, %d) on device 0" % (lo0, hi0) in r.stdout
    assert "shard 1: channels [%d, %d) on device 0" % (lo1, hi1) in r.stdout
    per = n // calls
    stride = int(r.stdout.split("stride ")[1].split()[0])
    bits = np.fromfile(f_bits, np.uint8).reshape(calls, Cn, stride)
    nb = np.fromfile(f_nb, np.int32).reshape(calls, Cn)
    if mode == "multibank-cs16":      # what the GPU computes on: x / 32768 of the rounded int16 (exact in binary32)
        q = np.clip(np.rint(iq.view(np.float32) * np.float32(32768.0)), -32768, 32767).astype(np.int16)
        iq = (q.astype(np.float32) * np.float32(1.0 / 32768.0)).view(np.complex64)
    cd = np.fromfile(str(f_bits) + ".cd", np.complex64).reshape(Cn, 1024)
    cdn = np.fromfile(str(f_bits) + ".cdn", np.int32)
    for c in range(Cn):
        o = oracle.Oracle()
        syms = []
        for k in range(calls):
            res = o.process(iq[c, k * per:(k + 1) * per])
            want = res["bits"]
            syms.append(res["sym"])
            assert nb[k, c] == want.size and np.array_equal(bits[k, c, :want.size], want), (c, k)
            if k == calls // 2:
                o.set_param(4, 0.02)
        # MultiBank::constellation: the plugin's constellation tap (src/main.cpp:85-89, :376-383) shard by shard = the last complete
        # 1024-symbol block of the oracle's symbol stream
        syms = np.concatenate(syms)
        done = syms.size // 1024
        assert cdn[c] == done and done >= 9
        assert np.array_equal(cd[c].view(np.uint32), syms[(done - 1) * 1024:done * 1024].view(np.uint32)), c


EXE4 = os.path.join(ROOT, "tests", "host", "test_config4")
CONFIG4_AMPS = (1.0, 0.37, 0.81, 0.052, 0.6, 0.23, 0.95, 0.11)          # test_config4.hip: amps[]


def _build_config4(pkg):
    pkg.build.build()
    drv = os.path.join(ROOT, "tests", "host", "test_config4.hip")
    mirror = os.path.join(PK, "host", "pi4dqpsk_gpu.cpp")
    deps = [drv, mirror, os.path.join(PK, "host", "pi4dqpsk_gpu.h"), os.path.join(PK, "host", "dsp_compat.h"),
            os.path.join(ROOT, "include", "tetra_demod.h")]
    if not os.path.exists(EXE4) or any(os.path.getmtime(d) > os.path.getmtime(EXE4) for d in deps):
        obj = EXE4 + ".mirror.o"
        # the product side stays plain C++ (g++); only the driver -- it generates its input on the GPU -- is HIP
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-c", mirror, "-o", obj], check=True)
        dobj = EXE4 + ".driver.o"
        subprocess.run([pkg.build.hipcc_path(), "--offload-arch=gfx950", "-std=c++17", "-O2", "-c", drv, "-o", dobj], check=True)
        subprocess.run([pkg.build.hipcc_path(), "--offload-arch=gfx950", "-pthread", dobj, obj, "-L", PK, "-ltetra_demod_hip",
                        "-Wl,-rpath," + PK, "-o", EXE4], check=True)
        os.remove(obj)
        os.remove(dobj)
    return EXE4


def config4_channel_input(base, c):
    """Channel c of test_config4.hip's bank, rebuilt bit for bit: base[c % B] x amp[(c / B) % 8] x j^((c / B) % 4)."""
    B = base.shape[0]
    k = c // B
    v = (base[c % B].view(np.float32) * np.float32(CONFIG4_AMPS[k % 8])).reshape(-1, 2)
    re, im = v[:, 0], v[:, 1]
    re, im = ((re, im), (-im, re), (-re, -im), (im, -re))[k & 3]
    out = np.empty((v.shape[0], 2), np.float32)
    out[:, 0], out[:, 1] = re, im
    return out.reshape(-1).view(np.complex64)


def test_config4_input_rule_is_exact():
    """The bank rule above uses only exact float operations besides one multiply per component (CPU check of the helper)."""
    rng = np.random.default_rng(3)
    base = (rng.standard_normal((4, 64)) + 1j * rng.standard_normal((4, 64))).astype(np.complex64)
    for c in range(4 * 8 * 2):
        k = c // 4
        want = base[c % 4].astype(np.complex128) * float(np.float32(CONFIG4_AMPS[k % 8])) * (1j ** (k & 3))
        got = config4_channel_input(base, c)
        assert np.allclose(got, want, rtol=1e-6, atol=0) and got.dtype == np.complex64
    assert np.array_equal(config4_channel_input(base, 1), config4_channel_input(base, 1 + 4 * 8))          # the period


@pytest.mark.gpu
def test_config4_32768_channels_eight_shards(pkg, oracle, synth, tmp_path):
    """BASELINE config 4 at full size: 32768 channels = 8 shards x 4096 (per-GPU channel ranges, no collective;
    /root/reference src/main.cpp:51: one chain per instance, any number of instances), 36000 samples per channel, ONE
    PI4DQPSKMultiBank, input generated on the GPU, processDevice.  A GPU per shard where the box has eight; on the one-GPU box
    all eight handles share it, each driven by its own thread on its own stream.
      * 80 channels -- both sides of every shard boundary, the bank's first and last channel, and others spread over all
        shards -- against the oracle on the bit-identical input: counts and every bit;
      * the first 512 channels (one period of the input rule): transmitted bits come back after lock;
      * ALL other channels: row and count identical to the channel one period earlier, which received the same samples."""
    exe = _build_config4(pkg)
    B, n, Cn, G = 64, 36000, 32768, 8
    base, txb, _ = synth.gen_batch(B, n, base_seed=4400, amp=1.0)
    f_base = tmp_path / "base.f32"
    np.ascontiguousarray(base).view(np.float32).tofile(f_base)
    P = B * 8
    edges = [pkg.shard.channel_range(Cn, G, g) for g in range(G)]
    assert all(hi - lo == 4096 for lo, hi in edges)
    oracle_ch = sorted({0, Cn - 1} | {lo - 1 for lo, _ in edges[1:]} | {lo for lo, _ in edges[1:]} |
                       {int(c) for c in np.random.default_rng(44).integers(0, Cn, 64)})
    assert len(oracle_ch) >= 64
    sel = list(range(P)) + [c for c in oracle_ch if c >= P]
    f_sel = tmp_path / "sel.txt"
    f_sel.write_text("\n".join(str(c) for c in sel))
    f_rows, f_nb = tmp_path / "rows.u8", tmp_path / "nb.i32"
    r = subprocess.run([exe, str(f_base), str(B), str(n), str(Cn), str(G), str(f_rows), str(f_nb), str(f_sel)],
                       timeout=600, capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    import torch
    ngpu = torch.cuda.device_count()
    for g, (lo, hi) in enumerate(edges):
        assert "shard %d: channels [%d, %d) on device %d" % (g, lo, hi, g % ngpu) in r.stdout
    assert "period_check %d mismatches 0" % P in r.stdout, r.stdout[-500:]
    stride = int(r.stdout.split("stride ")[1].split()[0])
    rows = np.fromfile(f_rows, np.uint8).reshape(len(sel), stride)
    nb = np.fromfile(f_nb, np.int32)
    assert nb.shape == (Cn,) and nb.min() > 2 * n // 2 - 400 and nb.max() <= stride
    row_of = {c: i for i, c in enumerate(sel)}
    for c in oracle_ch:
        want = oracle.Oracle().process(config4_channel_input(base, c))["bits"]
        assert nb[c] == want.size and np.array_equal(rows[row_of[c], :want.size], want), c
    errs = ncmp = 0
    for c in range(P):
        lag, e, m = synth.align_and_count_errors(rows[c][: nb[c]], txb[c % B], skip=3 * nb[c] // 4)
        errs += e
        ncmp += m
    assert ncmp > 8000 * P and errs <= 1e-3 * ncmp, (errs, ncmp)          # bench.py's known-answer bound (Es/N0 25 dB)
