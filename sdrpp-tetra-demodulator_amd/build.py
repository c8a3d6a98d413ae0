"""Builds libtetra_demod_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtetra_demod_hip.so")
SOURCES = ["tetra_demod.hip", "tetra_chan.hip", "tetra_resamp.hip", "tetra_burst_scan.hip", "tetra_lmac.hip", "tetra_burst_sync.hip", "tetra_rx.hip"]
DEPS = ["tetra_rx.hip", os.path.join("..", "..", "include", "tetra_rx.h"), "tetra_burst_sync.hip", "bsync_core.hpp", "demux_core.hpp", os.path.join("..", "..", "include", "tetra_burst_sync.h"), "tetra_demod.hip", "tetra_chan.hip", "chan_fft_core.hpp", "tetra_resamp.hip", "resamp_core.hpp", "tetra_burst_scan.hip", "tetra_lmac.hip", "lmac_core.hpp", os.path.join("..", "..", "include", "tetra_lmac.h"), os.path.join("..", "..", "include", "tetra_burst_scan.h"), "demod_core.hpp", "constellation_core.hpp", "design.hpp", "kernel_fused.hpp", "kernel_generic.hpp", "fll_asm.inc", "fll4_asm.inc", "fll16_asm.inc", "fll16l_asm.inc", "fll8l_asm.inc", "gen_fll_asm.py",
        os.path.join("..", "..", "include", "tetra_demod.h"), os.path.join("..", "..", "include", "tetra_chan.h")]

# -ffp-contract=off + correctly rounded sqrt: the arithmetic contract shared with the oracle.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared"]
# Per-source additions.  tetra_chan.hip: no SLP packing -- the mixed-radix FFT's complex arithmetic otherwise becomes v_pk_*_f32 plus
# ~400 v_mov swizzles per block, and packed f32 is no faster than two scalar ops on gfx950 (MI355X_MICROARCH.md); measured 38.5 vs
# 41 us per 12500 frames (profiles/r05/README.md).  The demodulator's sources keep exactly the flags their kernels were tuned with.
EXTRA_FLAGS = {"tetra_chan.hip": ["-fno-slp-vectorize"]}


def hipcc_path():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")
    return p


BUILD_ID_MARKER = b"TETRA_BUILD_ID="


def source_hash():
    """sha256 over every source the library is compiled from (file names + contents, DEPS order de-duplicated) and the
    compile flags.  The library carries the value it was built from (tetra_demod_build_id(), the string behind
    BUILD_ID_MARKER in the file): a library is current exactly when the two agree -- file times say nothing after a checkout
    or a copy to another machine."""
    import hashlib
    hsh = hashlib.sha256()
    seen = set()
    for d in DEPS:
        path = os.path.normpath(os.path.join(CSRC, d))
        if path in seen:
            continue
        seen.add(path)
        with open(path, "rb") as f:
            hsh.update(os.path.basename(path).encode() + b"\0" + f.read() + b"\0")
    hsh.update(" ".join(HIPCC_FLAGS).encode())
    for src in sorted(EXTRA_FLAGS):
        hsh.update((src + " " + " ".join(EXTRA_FLAGS[src])).encode())
    return hsh.hexdigest()


def lib_build_id(path=None):
    """The build id embedded in a built library, read from the file itself (no dlopen: a stale library must not end up in the
    process that is about to replace it).  None if the file is missing or carries none."""
    path = path or LIB
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    at = blob.find(BUILD_ID_MARKER)
    if at < 0:
        return None
    hexid = blob[at + len(BUILD_ID_MARKER): at + len(BUILD_ID_MARKER) + 64]
    try:
        return hexid.decode("ascii") if len(hexid) == 64 and int(hexid, 16) >= 0 else None
    except ValueError:
        return None


def is_stale():
    return lib_build_id() != source_hash()


def build(force=False, verbose=False):
    """Compile the library if missing or built from other sources than the tree holds (build id, see source_hash).  Returns the .so path.  Serialised across processes by a
    lock file (the ranks of a multi-GPU run all import the package at once: one builds, the others find the result), and the
    library appears atomically (compiled next to its final name, then renamed)."""
    if not (force or is_stale()):
        return LIB
    import fcntl
    import sys
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or is_stale():
                # fll_asm.inc is generated (and committed): refuse to build from one that is not what the generator emits
                subprocess.run([sys.executable, os.path.join(CSRC, "gen_fll_asm.py"), "--check"], check=True, stdout=subprocess.DEVNULL)
                tmp = LIB + ".tmp.%d" % os.getpid()
                objs = []
                try:
                    # one object per source (its own flags), compiled side by side, then one link
                    base = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-DTETRA_BUILD_ID=\"%s\"" % source_hash(), "-c"]
                    procs = []
                    for src in SOURCES:
                        obj = "%s.%s.o" % (tmp, src)
                        objs.append(obj)
                        cmd = [hipcc_path()] + base + EXTRA_FLAGS.get(src, []) + ["-o", obj, os.path.join(CSRC, src)]
                        if verbose:
                            print(" ".join(cmd))
                        procs.append((cmd, subprocess.Popen(cmd)))
                    for cmd, pr in procs:
                        if pr.wait() != 0:
                            raise subprocess.CalledProcessError(pr.returncode, cmd)
                    link = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp] + objs
                    if verbose:
                        print(" ".join(link))
                    subprocess.run(link, check=True)
                    os.replace(tmp, LIB)
                finally:
                    for f in objs + [tmp]:
                        if os.path.exists(f):
                            os.remove(f)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB
