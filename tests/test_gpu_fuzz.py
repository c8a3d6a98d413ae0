"""profiles/fuzz_parity.py as part of the suite: a fixed-seed slice of each of its three modes on the GPU (the long runs are
recorded under profiles/r03/), and its oracle-side mechanics on the CPU."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzz(pkg, oracle):
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "profiles", "fuzz_parity.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _run(fuzz, case, n, seed):
    rng = np.random.default_rng(seed)
    stats = dict(cases=0, refused=0, channel_calls=0, bits=0)
    for _ in range(n):
        f = case(rng, stats)
        assert f is None, f
    return stats


def test_fuzzer_mechanics_without_a_gpu(fuzz):
    fuzz.DRY = True
    try:
        stats = _run(fuzz, fuzz.one_case, 6, 3)
    finally:
        fuzz.DRY = False
    assert stats["bits"] > 10000


@pytest.mark.gpu
def test_chain_fuzz_slice(fuzz):
    """Random rates, tap counts, loop constants, shapes, layouts, outputs, statistic, quirks + resets, setter mid-stream,
    degenerate channels: bits, counts, symbols, statistic, state == oracle."""
    stats = _run(fuzz, fuzz.one_case, 120, 2024)
    assert stats["channel_calls"] > 5000


@pytest.mark.gpu
def test_launch_plan_fuzz_slice(fuzz):
    """Channel counts 1 ... 20000 on the automatic launch plan, two calls with carried state == oracle."""
    stats = _run(fuzz, fuzz.plan_case, 150, 2025)
    assert stats["channel_calls"] > 50000


@pytest.mark.gpu
def test_async_path_fuzz_slice(fuzz):
    """Call lengths around the asynchronous path's chunk boundaries, float / int16 / int8, both layouts, two calls in flight."""
    stats = _run(fuzz, fuzz.async_case, 8, 2026)
    assert stats["bits"] > 100000
