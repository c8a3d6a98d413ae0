"""bench.py's N > 1 path on real hardware: `python bench.py --gpus 2` by itself starts two ranks (here both mapped onto the
one GPU of the box), once over gloo and once over RCCL; on a box with fewer GPUs than ranks the line is marked functional_only
(value null, n_gpus = the GPUs that ran) and the known-answer check is green."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def _run(backend):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--backend", backend]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)


def _line(r):
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return json.loads(lines[0])


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_bench_gpus_2_runs_two_ranks(backend):
    """`python bench.py --gpus 2 ...` alone.  With a GPU per rank the default backend is RCCL; on a box with fewer GPUs than ranks
    (both ranks on the one GPU here) RCCL would refuse the communicator ("Duplicate GPU detected"), so bench.py lines the ranks up
    over gloo and says so in the line -- either way two ranks run; a folded run can not be mistaken for a 2-GPU measurement."""
    import torch
    r = _run(backend)
    d = _line(r)
    os.makedirs(OUT, exist_ok=True)
    json.dump(d, open(os.path.join(OUT, "bench_gpus2_%s.json" % backend), "w"))
    assert d["n_ranks"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["ramp_steps"] == 8
    ngpu = torch.cuda.device_count()
    assert d["gpus_physical"] == ngpu and d["n_gpus"] == min(2, ngpu) and d["functional_only"] == (ngpu < 2)
    value = d["value"]
    if ngpu < 2:
        assert d["value"] is None and "NOT an" in d["functional_note"]
        value = d["value_functional"]
    if backend == "nccl" and torch.cuda.device_count() >= 2:
        assert d["dist_backend"] == "nccl" and d["rccl_world_size"] == 2
    else:
        assert d["dist_backend"] == "gloo" and d["rccl_world_size"] is None
        assert ("dist_backend_note" in d) == (backend == "nccl")
    assert d["check"]["bit_errors"] <= 1e-3 * d["check"]["bits_compared_last_quarter"]
    # two ranks of 4096 channels each: the whole-job value counts both
    assert abs(value - 2 * 4096 * 36000 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * value


def test_bench_gpus_8_folded_onto_the_box(tmp_path):
    """the driver's SCALE run is `bench.py --gpus 8` on an 8-GPU node that no round has had.  Here the same
    command starts EIGHT ranks (512 channels each so that eight handles, inputs and bit rows fit beside each other on one GPU),
    folded onto the GPUs the box has: rendezvous, eight communicators' worth of barriers and the MAX / SUM reductions, eight
    known-answer checks -- every rank's own channel range against its own transmitted bits -- and a line nobody can mistake for
    an 8-GPU measurement."""
    import torch
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--channels", "512",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    d = _line(r)
    os.makedirs(OUT, exist_ok=True)
    json.dump(d, open(os.path.join(OUT, "bench_gpus8_folded.json"), "w"))
    ngpu = torch.cuda.device_count()
    assert d["n_ranks"] == 8 and d["gpus_physical"] == ngpu and d["n_gpus"] == min(8, ngpu) and d["scaling"] == "weak"
    assert d["functional_only"] == (ngpu < 8)
    if ngpu < 8:
        assert d["value"] is None and d["value_functional"] > 0 and "NOT an 8-GPU measurement" in d["functional_note"]
        assert d["dist_backend"] == "gloo" and "RCCL needs one GPU per rank" in d["dist_backend_note"]
    ck = d["check"]
    assert ck["ranks_checked"] == 8 and ck["ranks_green"] == 8 and ck["channels_checked"] == 8 * 256          # bench.CHECK_CHANNELS of every rank
    assert ck["bit_errors"] <= 1e-3 * ck["bits_compared_last_quarter"] and ck["bits_compared_last_quarter"] > 8 * 32 * 8000
    assert d["config"]["channels_per_gpu"] == 512


def test_bench_rccl_half_of_the_rank_path_on_one_gpu():
    """RCCL refuses several ranks on one device, so on a one-GPU box its half of bench.py's N > 1 path -- communicator set-up
    with device_id, barrier, MAX all-reduce of the elapsed time on a device tensor, tear-down -- runs as a ONE-rank group
    (--force-dist); the two-rank half runs over gloo above."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-host-path", "--no-large-batch", "--no-config5", "--no-time-major", "--force-dist", "--backend", "nccl"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    d = _line(r)
    os.makedirs(OUT, exist_ok=True)
    json.dump(d, open(os.path.join(OUT, "bench_force_dist_nccl.json"), "w"))
    assert d["n_gpus"] == 1 and d["dist_backend"] == "nccl" and d["rccl_world_size"] == 1
    assert d["check"]["bit_errors"] <= 1e-3 * d["check"]["bits_compared_last_quarter"]


def test_bench_under_the_driver_s_own_launcher():
    """The driver's command line for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- bench.py then runs as ONE of the ranks (it must not start ranks of its own)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    d = _line(r)
    import torch
    assert d["n_ranks"] == 2 and d["n_gpus"] == min(2, torch.cuda.device_count()) and d["dist_backend"] == "gloo"
    assert d["functional_only"] == (torch.cuda.device_count() < 2)
    # 256 channels per rank from their own seeds at 25 dB: a few true channel errors among millions of bits (bench.py's own bound)
    assert d["check"]["bit_errors"] <= 1e-3 * d["check"]["bits_compared_last_quarter"]


def test_chain_leg_known_answer_at_full_size():
    """The receive chain at BASELINE's size through the one property that does not need the reference beside it: 4096 channels x 36000
    samples of CODED downlinks (the reference's encoder chain restated in synth.gen_downlink, pinned on the reference's own encoder
    by tests/test_synth_tx.py) -> tetra_rx -> every one of the ~651 000 blocks of the checked call has a good CRC, every block of
    the 64 distinct streams carries exactly the type-1 bits sent in the slot its TDMA label names, every channel is locked and reads
    its cell -- on two streams and on one (`bench.py --chain-only` fails the run otherwise; the counters are asserted here too)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--chain-only"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    os.makedirs(OUT, exist_ok=True)
    json.dump(d, open(os.path.join(OUT, "bench_chain_only.json"), "w"))
    ck = d["check"]
    assert ck["channels"] == 4096 and ck["channels_locked"] == 4096 and ck["cells_read"] == 4096
    total = 0
    for nm, c in ck["blocks"].items():
        assert c["rows"] == c["crc_good"] > 60000 and c["checked_rows"] == c["type1_bits_and_tdma_slot_exact"] > 1000, (nm, c)
        total += c["rows"]
    assert total > 600000 and d["rows_per_kind"]["schf"] == ck["blocks"]["schf"]["rows"]
    assert 3.5 < d["two_streams_ms_per_second"] < d["one_stream_ms_per_second"] * 1.02 < 6.0        # sanity: not a timing claim
    assert set(d["stages_one_stream"]) == {"demodulator", "burst_sync", "frame_lists_sb1_decode_track", "other_kinds_decode_label"}
