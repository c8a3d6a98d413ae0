"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the CPU checkers (oracle/*.c) and of the host-side glue
(sdrpp-tetra-demodulator_amd/host/).  The drivers under tests/san/ are built with
-fsanitize=address,undefined -fno-sanitize-recover=all, so any report is a non-zero exit."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "san")
PK = os.path.join(ROOT, "sdrpp-tetra-demodulator_amd")
FLAGS = ["-g", "-O1", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
           LSAN_OPTIONS="suppressions=" + os.path.join(SAN, "lsan.supp") + ":print_suppressions=0")


def _stale(exe, deps):
    return not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps)


def test_oracles_are_clean_under_asan_and_ubsan():
    exe = os.path.join(SAN, "san_oracle")
    srcs = [os.path.join(SAN, "san_oracle.c")] + [os.path.join(ROOT, "oracle", f)
                                                  for f in ("tetra_oracle.c", "burst_sync_oracle.c", "chan_oracle.c")]
    if _stale(exe, srcs + [os.path.join(ROOT, "oracle", "tetra_oracle.h")]):
        # same arithmetic flags as oracle/Makefile; OpenMP stays on (the batch driver's threads are part of what is checked)
        subprocess.run(["gcc", "-std=c11", "-ffp-contract=off", "-mfma", "-mavx2", "-fopenmp", "-Wall"] + FLAGS + srcs +
                       ["-lm", "-o", exe], check=True)
    r = subprocess.run([exe], env=ENV, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "san_oracle: ok" in r.stdout, r.stderr[-3000:]


def test_kernel_thread_code_is_clean_under_asan_and_ubsan():
    """The bit-moving thread-level code of the demultiplexer kernels (csrc/demux_core.hpp), of the constellation tap
    (csrc/constellation_core.hpp) and -- round 6 -- the lower-MAC decoder's lane code (csrc/lmac_core.hpp: byte rows through both front
    ends, blocks cut straight from packed frames of every burst type), host builds, on exact-size heap buffers."""
    exe = os.path.join(SAN, "san_emul")
    emul = os.path.join(ROOT, "tests", "emul")
    srcs = [os.path.join(SAN, "san_emul.cpp"), os.path.join(emul, "bsync_emul.cpp"), os.path.join(emul, "lmac_emul.cpp")]
    deps = srcs + [os.path.join(PK, "csrc", f) for f in ("demux_core.hpp", "bsync_core.hpp", "constellation_core.hpp", "lmac_core.hpp")]
    if _stale(exe, deps):
        subprocess.run(["g++", "-std=c++17", "-Wall"] + FLAGS + srcs + ["-o", exe], check=True)
    r = subprocess.run([exe], env=ENV, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "san_emul: ok" in r.stdout, (r.stdout[-300:], r.stderr[-3000:])


def _build_host(pkg):
    pkg.build.build()
    exe = os.path.join(SAN, "san_host")
    srcs = [os.path.join(SAN, "san_host.cpp")] + [os.path.join(PK, "host", f) for f in ("pi4dqpsk_gpu.cpp", "dqpsk_sym_extr_gpu.cpp",
                                                                                         "bit_unpacker_gpu.cpp")]
    deps = srcs + [os.path.join(PK, "host", f) for f in ("pi4dqpsk_gpu.h", "dqpsk_sym_extr_gpu.h", "bit_unpacker_gpu.h", "dsp_compat.h")] + \
        [os.path.join(ROOT, "include", "tetra_demod.h")]
    if _stale(exe, deps):
        subprocess.run(["g++", "-std=c++17", "-Wall", "-pthread"] + FLAGS + srcs +
                       ["-L", PK, "-ltetra_demod_hip", "-Wl,-rpath," + PK, "-o", exe], check=True)
    return exe


def test_host_glue_is_clean_under_asan_and_ubsan(pkg):
    """stream / block / Processor threading and the GPU classes' error paths; no GPU needed."""
    exe = _build_host(pkg)
    r = subprocess.run([exe], env=ENV, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "san_host: ok" in r.stdout, r.stderr[-3000:]


@pytest.mark.gpu
def test_host_glue_streams_through_the_gpu_under_asan_and_ubsan(pkg):
    """The same binary with real PI4DQPSK / PI4DQPSKBank / PI4DQPSKMultiBank traffic.  What the ROCm runtime keeps until
    exit is suppressed by library name (tests/san/lsan.supp); leaks of our own code still fail the run."""
    exe = _build_host(pkg)
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=1:protect_shadow_gap=0")
    r = subprocess.run([exe, "gpu"], env=env, capture_output=True, text=True, timeout=90)
    assert r.returncode == 0 and "san_host: ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def _build_rx_bank(pkg):
    pkg.build.build()
    exe = os.path.join(SAN, "san_rx_bank")
    srcs = [os.path.join(SAN, "san_rx_bank.cpp")]
    deps = srcs + [os.path.join(PK, "host", "tetra_rx_bank.h")] + [os.path.join(ROOT, "include", f) for f in ("tetra_rx.h", "tetra_lmac.h", "tetra_demod.h")]
    if _stale(exe, deps):
        subprocess.run(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PK, "host")] + FLAGS + srcs +
                       ["-L", PK, "-ltetra_demod_hip", "-Wl,-rpath," + PK, "-o", exe], check=True)
    return exe


def test_rx_bank_error_paths_are_clean_under_asan_and_ubsan(pkg):
    """host/tetra_rx_bank.h (the C++ face of include/tetra_rx.h, beside PI4DQPSKBank): bad configurations and a bank without a handle
    are statuses; no GPU needed."""
    r = subprocess.run([_build_rx_bank(pkg)], env=ENV, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "san_rx_bank: ok" in r.stdout, r.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [0, 2, 3])
def test_rx_bank_streams_like_rx_chain_under_asan_and_ubsan(pkg, synth, tmp_path, shards):
    """TetraRxBank (shards = 0) / TetraRxMultiBank (the bank's channels split into that many channel ranges, each range a handle of its
    own -- all on device 0 here; on a node one per GPU: the chain shards like the demodulator, no exchange) fed three blocks of coded downlinks from host memory, fetching the previous call's blocks while the next one runs:
    every block of every kind (labels + type-1 bits) and the cell states equal what the Python RxChain returns for the same stream,
    and the sanitizers stay silent (ROCm's own allocations suppressed by library name)."""
    import numpy as np
    from tests.test_rx import _downlink_batch
    R = pkg.rx_binding
    Cn, calls, nslots = 5, 3, 36
    N = nslots * 510 // calls
    cells, tx, iq = _downlink_batch(synth, Cn, nslots, N * calls, 8800)
    blocks_file, out_file = tmp_path / "iq.bin", tmp_path / "out.bin"
    with open(blocks_file, "wb") as f:
        for k in range(calls):
            f.write(np.ascontiguousarray(iq[:, k * N:(k + 1) * N]).astype(np.complex64).tobytes())
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=1:protect_shadow_gap=0")
    r = subprocess.run([_build_rx_bank(pkg), "gpu", str(Cn), str(N), str(calls), str(blocks_file), str(out_file)] + ([str(shards)] if shards else []),
                       env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "san_rx_bank: ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    # the same stream through the Python handle
    rx = pkg.RxChain(Cn, N)
    want = []
    for k in range(calls):
        rx.process(np.ascontiguousarray(iq[:, k * N:(k + 1) * N]))
        rx.wait()
        want.append([rx.fetch(kind) for kind in range(R.N_KINDS)])
    want_cells = [tuple(getattr(c, f) for f, _ in R.CellState._fields_) for c in rx.cells()]
    rx.close()
    raw = open(out_file, "rb").read()
    at, total = 0, 0
    for k in range(calls):
        for kind in range(R.N_KINDS):
            hk, n, nb = np.frombuffer(raw, np.int32, 3, at)
            at += 12
            info = np.frombuffer(raw, R.BLOCK_DTYPE, n, at)
            at += n * R.BLOCK_DTYPE.itemsize
            t1 = np.frombuffer(raw, np.uint8, n * nb, at).reshape(n, nb)
            at += n * nb
            wb, wt = want[k][kind]
            assert hk == kind and nb == R.type1_bits(kind) and n == len(wb), (k, kind, n, len(wb))
            assert info.tobytes() == wb.tobytes() and np.array_equal(t1, wt), (k, kind)
            total += n
    got_cells = np.frombuffer(raw, np.uint32, 10 * Cn, at).reshape(Cn, 10)
    assert [tuple(int(v) for v in row) for row in got_cells] == want_cells and total > 200
