"""bench.py pieces that can be checked without a GPU: the counter-traffic staleness rule and the constants of the line."""
import json
import os
import shutil

import bench


def test_counter_traffic_is_refused_for_other_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed rocprofv3 measurement stamped with the sha256 of the kernel sources it was
    taken on; for any other sources bench.py reports null and says why."""
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    csrc = tmp_path / "sdrpp-tetra-demodulator_amd" / "csrc"
    os.makedirs(csrc)
    real = os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "sdrpp-tetra-demodulator_amd", "csrc")
    for name in bench.KERNEL_SOURCES:
        shutil.copy(os.path.join(real, name), csrc / name)
    h = bench.kernel_source_hash()
    path = tmp_path / "profiles" / "pmc_traffic_fused_4096x36000.json"
    json.dump({"traffic_bytes_per_launch": 1.25e9, "kernel_source_sha256": h, "issue": {"x": 1}}, open(path, "w"))
    t, issue, unit = bench.pmc_traffic("fused", 4096, 36000)
    assert t == 1.25e9 and issue == {"x": 1} and "pmc_traffic_fused_4096x36000.json" in unit
    with open(csrc / bench.KERNEL_SOURCES[0], "a") as f:          # the kernel changes: the measurement no longer applies
        f.write("\n// edited\n")
    t, issue, unit = bench.pmc_traffic("fused", 4096, 36000)
    assert t is None and issue is None and unit.startswith("stale")
    t, issue, unit = bench.pmc_traffic("fused", 256, 36000)          # no measurement for that workload at all
    assert t is None and "no counter measurement" in unit


def test_committed_counter_traffic_matches_the_committed_kernel():
    """What is in the tree is consistent: the committed measurement was taken on the committed kernel sources."""
    for channels in (4096, 8192):          # the bench workload (16-channel workgroups) and the large batch (32-channel ones)
        t, _, unit = bench.pmc_traffic("fused", channels, 36000)
        assert t is not None, unit
        assert 1.0 <= t / (bench.ALGO_BYTES_PER_SAMPLE * channels * 36000) < 1.05


def test_line_constants():
    assert bench.ALGO_BYTES_PER_SAMPLE == 9.0 and bench.HBM_PEAK_GBS == 8000.0 and bench.VALU_PEAK_TFLOPS == 157.3
    assert bench.CHANNELS_PER_GPU == 4096 and bench.SAMPLES == 36000
    # the flop count's itemisation adds up
    assert 9 + 36 + 520 + 12 + 260 + (96 + 12 + 72 + 16) / 2 == bench.FLOP_PER_SAMPLE


def test_gpus_flag_launches_that_many_ranks(monkeypatch):
    """`python bench.py --gpus N` (N > 1, no launcher around it) starts N ranks of the same command under torch.distributed.run
    on 127.0.0.1 -- the driver's plain command line must not record N copies of the 1-GPU number."""
    import subprocess
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    assert bench.launch_ranks(4, ["--gpus", "4", "--steps", "3"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    # the launcher picks the rendezvous port itself (no find-a-port-then-bind race), on 127.0.0.1
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"
    assert "--master-port" not in cmd
    assert cmd[-5:] == [os.path.abspath(bench.__file__), "--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_rank_count_must_equal_gpus_flag():
    """Under a launcher the world size has to be what --gpus says, or the line would claim a GPU count it did not run on."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.abspath(bench.__file__))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "is running as one of 2 rank(s)" in r.stderr


def test_default_line_carries_config5_at_the_plugins_rate_and_the_receive_chain():
    """What the driver's plain `python bench.py` run reports beside the metric: config 5 through the 18/25
    resampler at the plugin's 36 ksps by default, the receive chain (include/tetra_rx.h) unless --no-chain, per-channel seeds and a
    known-answer check over >= 256 channels.  (Source-level pins: the line itself needs the GPU; tests/test_bench_ranks.py runs it.)"""
    import inspect
    src = inspect.getsource(bench.main)
    assert 'out["chain"] = informational(receive_chain, ' in src and "args.no_chain" in src and 'out["config5"] = informational(wideband_config5, ' in src
    # an informational leg that fails (its own check included) is reported in place, the metric line still comes out
    assert bench.informational(lambda: (_ for _ in ()).throw(SystemExit("check failed"))) == {"error": "SystemExit: check failed"}
    assert bench.informational(lambda x: {"ok": x}, 3) == {"ok": 3}
    assert bench.CHECK_CHANNELS >= 256 and "hash_bits(int(seeds[c])" in src
    c5 = inspect.getsource(bench.wideband_config5)
    assert "pkg.Resampler(M, 18, 25, 16" in c5 and '"resampler"' in c5 and "config5_rate" in c5
    chain = inspect.getsource(bench.receive_chain)
    for key in ("two_streams", "one_stream", "stages_one_stream", "crc_good", "type1_bits_and_tdma_slot_exact", "cells_read", "channels_locked"):
        assert key in chain, key
    ap = inspect.getsource(bench.main)
    assert '"--config5-rate", type=int, default=36000' in ap


def test_hashed_channel_streams_are_the_same_on_cpu_and_in_torch():
    """bench.py's input: every channel from its own seed.  synth.hash_bits / hash_params (numpy) == synth_gpu's torch arithmetic, and
    the torch modulator gives synth.gen_channel's samples (run on the CPU device here)."""
    import numpy as np
    import torch
    import tetra_amd
    pkg = tetra_amd.pkg
    dev = torch.device("cpu")
    iq, seeds = pkg.synth_gpu.gen_bank(torch, dev, 6, 1500, 20260000, esn0_db=None)
    assert list(seeds) == list(range(20260000, 20260006))
    for c in range(6):
        sd = int(seeds[c])
        bits = pkg.synth.hash_bits(sd, pkg.synth.needed_bits(1500))
        assert 0.4 < bits.mean() < 0.6
        ref = pkg.synth.gen_channel(1500, 0, bits=bits, esn0_db=None, **pkg.synth.hash_params(sd))[0]
        assert np.abs(iq[c].numpy() - ref).max() <= 1e-6 * np.abs(ref).max()
    b = pkg.synth_gpu.hash_bits(torch, dev, seeds, 300).numpy()
    assert all(np.array_equal(b[c], pkg.synth.hash_bits(int(seeds[c]), 300)) for c in range(6))
    assert not np.array_equal(b[0], b[1])
