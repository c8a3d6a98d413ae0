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
