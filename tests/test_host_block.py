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
    srcs = [os.path.join(ROOT, "tests", "host", "test_block.cpp"), os.path.join(PK, "host", "pi4dqpsk_gpu.cpp")]
    deps = srcs + [os.path.join(PK, "host", "pi4dqpsk_gpu.h"), os.path.join(PK, "host", "dsp_compat.h"),
                   os.path.join(ROOT, "include", "tetra_demod.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        # libamdhip64 only for the test driver's own device buffers ("multibank-device"); the mirror itself needs just the C ABI
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + srcs +
                       ["-L", PK, "-ltetra_demod_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + PK, "-Wl,-rpath,/opt/rocm/lib",
                        "-o", EXE], check=True)
    return EXE


def test_block_mirror_builds_and_links(pkg):
    exe = _build(pkg)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "abi 3" in out and "default taps 65" in out


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
    for c in range(Cn):
        o = oracle.Oracle()
        for k in range(calls):
            want = o.process(iq[c, k * per:(k + 1) * per])["bits"]
            assert nb[k, c] == want.size and np.array_equal(bits[k, c, :want.size], want), (c, k)
            if k == calls // 2:
                o.set_param(4, 0.02)
