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
