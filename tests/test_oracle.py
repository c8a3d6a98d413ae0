"""CPU tests of the oracle (oracle/tetra_oracle.c): design constants, known answers, invariances.

The reference ships no tests or vectors for this path and cannot be built here, so the oracle is
'parity unpinned' (oracle/tetra_oracle.h).  What pins it instead:
  * the derived constants the survey probed from the reference code,
  * known-answer runs: transmitted bits come back, with the ETSI bit<->phase map of the
    reference's own tables (src/decoder/src/phy/tetra_burst.c:99-117) and its training sequences,
  * frozen outputs under tests/golden/ (regression pins of this restatement, NOT reference-derived).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_design_constants(oracle):
    o = oracle.Oracle()
    t = o.tab
    # SURVEY.md Appendix B.7 (probe of the reference code)
    assert abs(t.fll_beta - 1.42783e-4) < 1e-9 and t.fll_alpha == 0.0
    assert abs(t.costas_alpha - 0.0278871) < 1e-7 and abs(t.costas_beta - 3.94383e-4) < 1e-9
    assert abs(t.tr_alpha - 0.0176028) < 1e-7 and abs(t.tr_beta - 1.56359e-4) < 1e-9
    assert abs(t.tr_min_freq - 1.96) < 1e-6 and abs(t.tr_max_freq - 2.04) < 1e-6
    rrc = o.rrc_taps()
    # symmetric, sum 1.0001, peak h[32] = 0.547817, energy 0.5
    assert np.array_equal(rrc, rrc[::-1])
    assert abs(rrc.sum() - 1.0001) < 1e-4 and abs(rrc[32] - 0.547817) < 1e-6 and abs((rrc ** 2).sum() - 0.5) < 1e-4
    a, b = o.bandedge_taps()
    assert np.array_equal(a, a[::-1]) and np.array_equal(b, -b[::-1])  # conjugate-pair structure
    H = np.fft.fftshift(np.fft.fft((a + 1j * b)[::-1], 8192))
    f = np.fft.fftshift(np.fft.fftfreq(8192, 1 / 36000.0))
    cen = (f * abs(H) ** 2).sum() / (abs(H) ** 2).sum()
    assert -13000 < cen < -11000  # lower band-edge filter sits at ~ -(1+alpha)*Rs/2 = -12.15 kHz (survey: ~ -11.7 kHz peak)
    bank = o.interp_bank()
    assert np.allclose(bank.sum(1), 1.0, atol=1e-4) and bank[0].argmax() == 3 and bank[127].argmax() == 4


def test_sincos_accuracy(oracle):
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for x in np.linspace(-7.0, 7.0, 20001).astype(np.float32):
        oracle.lib().tetra_oracle_sincosf(C.c_float(float(x)), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(np.float64(x))), abs(c.value - np.cos(np.float64(x))))
    assert worst < 2.0e-7  # ~2.5 ulp at 1.0: far inside the symbol tolerance vs libm cosf/sinf


def test_known_answer_lock_and_lag(oracle, synth):
    """Same scenario as the survey's probe of the reference: 40060 samples -> 20031 symbols, lock, no errors."""
    N = 40060
    iq, txb, _ = synth.gen_channel(N, 7, cfo=0.03, tau=5 / 16.0, amp=0.2, esn0_db=25.0)
    r = oracle.Oracle().process(iq)
    assert len(r["sym"]) == 20031 and len(r["bits"]) == 40062
    lag, err, n = synth.align_and_count_errors(r["bits"], txb, skip=len(r["bits"]) // 2)
    assert err == 0 and n > 19000
    assert abs(np.abs(r["sym"][10000:]).mean() - 1.0) < 0.05   # AGC: |sym| ~ 1


def test_etsi_phase_map_noise_free(oracle, synth):
    """Each dibit value maps to its ETSI phase step: feed a stream of one repeated dibit (after a random
    preamble for lock) and read it back."""
    rng = np.random.default_rng(3)
    pre = rng.integers(0, 2, 6000, dtype=np.uint8)
    for d in range(4):
        tail = np.tile(np.array([(d >> 1) & 1, d & 1], np.uint8), 1200)
        bits = np.concatenate([pre, tail])
        N = bits.size - 100
        iq, _, _ = synth.gen_channel(N, 11, cfo=0.01, tau=0.4, amp=0.5, esn0_db=None, bits=bits)
        rx = oracle.Oracle().process(iq)["bits"]
        lag, err, n = synth.align_and_count_errors(rx, bits, skip=rx.size - 1500)
        assert err == 0 and n >= 1000, (d, lag, err, n)


def test_training_sequence_found_at_slot_spacing(oracle, synth):
    """Known-answer with the reference's protocol constants: 510-bit slots carrying the normal training
    sequence 1 at bit 244 (ETSI EN 300 392-2 9.4.4.3.2; src/decoder/src/phy/tetra_burst.c:61, same bits
    in src/main.cpp:457-468) are demodulated and the sequence is found every 510 bits."""
    with open(os.path.join(GOLDEN, "etsi_training_sequences.json")) as f:
        ts = json.load(f)
    n_seq = np.array(ts["normal_1"], np.uint8)
    rng = np.random.default_rng(5)
    slots = []
    for _ in range(24):
        s = rng.integers(0, 2, 510, dtype=np.uint8)
        s[244:244 + 22] = n_seq
        slots.append(s)
    bits = np.concatenate(slots)
    N = bits.size - 64
    iq, _, _ = synth.gen_channel(N, 21, cfo=-0.02, tau=1.3, amp=0.3, esn0_db=25.0, bits=bits)
    rx = oracle.Oracle().process(iq)["bits"]
    hits = [i for i in range(rx.size - 22) if np.array_equal(rx[i:i + 22], n_seq)]
    hits = [h for h in hits if h > 4000]
    good = [h for h in hits if any(abs((h - g) % 510) == 0 for g in hits[:1])]
    assert len(good) >= 14, hits
    assert all((good[i + 1] - good[i]) % 510 == 0 for i in range(len(good) - 1))


def test_reference_built_bursts_lock_on_the_oracle(oracle, synth, ref):
    """Reference-anchored known answer: bursts from the reference's OWN builders (tetra_burst.c:171-269, via
    oracle/_ref) -> IQ -> oracle chain -> the reference's OWN tetra_find_train_seq finds sync bursts every 4 slots
    and normal bursts at 510-bit spacing in the demodulated stream."""
    rng = np.random.default_rng(17)
    nslots = 40
    slots = []
    for s in range(nslots):
        if s % 4 == 0:
            slots.append(ref.build_sync_burst(rng.integers(0, 2, 120), rng.integers(0, 2, 30), rng.integers(0, 2, 216)))
        else:
            slots.append(ref.build_norm_burst(rng.integers(0, 2, 216), rng.integers(0, 2, 30), rng.integers(0, 2, 216), s % 2))
    tx = np.concatenate(slots)
    N = nslots * 510 - 200
    iq, _, _ = synth.gen_channel(N, 5, bits=tx)
    rx = oracle.Oracle().process(iq)["bits"]
    hits, pos = [], 6000
    while pos < rx.size - 600:
        t, o = ref.find_train_seq(rx[pos:], int(rx.size - pos - 64))
        if t < 0:
            break
        hits.append((t, pos + o))
        pos += o + 60
    sync = [o for t, o in hits if t == ref.TRAIN_SYNC]
    norm = [o for t, o in hits if t in (ref.TRAIN_NORM_1, ref.TRAIN_NORM_2)]
    assert len(hits) >= 25 and len(sync) >= 5
    assert all((b - a) % (4 * 510) == 0 for a, b in zip(sync, sync[1:]))
    assert all((b - a) % 510 == 0 for a, b in zip(norm, norm[1:]))
    # the demodulated slot payload equals the transmitted burst bit for bit (constant lag)
    lag = sync[-1] - 214 - ((sync[-1] - 214) // 510) * 510
    seg = rx[sync[-1] - 214: sync[-1] - 214 + 510]
    k = [i for i in range(nslots) if np.array_equal(seg, slots[i])]
    assert len(k) == 1 and k[0] % 4 == 0, lag


def test_chunk_invariance(oracle, synth):
    N = 9000
    iq, _, _ = synth.gen_channel(N, 5)
    ref = oracle.Oracle().process(iq)
    for ch in (1, 7, 180, 4096):
        o = oracle.Oracle()
        bb, ss = [], []
        for pos in range(0, N, ch):
            r = o.process(iq[pos:pos + ch])
            bb.append(r["bits"])
            ss.append(r["sym"])
        assert np.array_equal(np.concatenate(bb), ref["bits"])
        assert np.array_equal(np.concatenate(ss).view(np.uint32), ref["sym"].view(np.uint32))


def test_batch_driver_matches_single(oracle, synth):
    iq, _, _ = synth.gen_batch(6, 3000, base_seed=8)
    bits, nb, sym, _ = oracle.process_batch(iq, chunk=1000, threads=2, want_sym=True)
    for c in range(6):
        r = oracle.Oracle().process(iq[c])
        assert nb[c] == r["bits"].size and np.array_equal(bits[c][:nb[c]], r["bits"])
        assert np.array_equal(sym[c][:nb[c] // 2].view(np.uint32), r["sym"].view(np.uint32))


def test_empty_input(oracle):
    r = oracle.Oracle().process(np.zeros(0, np.complex64))
    assert r["bits"].size == 0


@pytest.mark.parametrize("name", ["golden_c3_n6000"])
def test_golden_fixture(oracle, name):
    """Frozen oracle outputs (tests/golden/make_golden.py): guards the arithmetic contract against drift."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    iq = g["iq"]
    for c in range(iq.shape[0]):
        o = oracle.Oracle()
        r = o.process(iq[c])
        nb = int(g["n_bits"][c])
        assert r["bits"].size == nb and np.array_equal(r["bits"], g["bits"][c][:nb])
        assert np.array_equal(r["sym"].view(np.uint32), g["sym"][c][:nb // 2].view(np.uint32))
        assert np.float32(o.st.fll_freq).view(np.uint32) == g["fll_freq"][c].view(np.uint32)
        assert np.float32(o.st.omega).view(np.uint32) == g["omega"][c].view(np.uint32)


def test_fast_port_decides_like_the_oracle(oracle, synth):
    """oracle/tetra_fast.c (bench.py's "port-fast" CPU baseline: -O3 -march=native -ffast-math, independent accumulators) is
    the same chain: same symbol counts, and once the loops have locked every bit equals the oracle's and the transmitted one."""
    iq, txb, _ = synth.gen_batch(6, 24000, base_seed=4242)
    b0, n0, _, st0 = oracle.process_batch(iq, threads=2)
    b1, n1, st1 = oracle.fast_process_batch(iq, chunk=5000, threads=2)
    assert np.array_equal(n0, n1)
    for c in range(6):
        h = n0[c] // 2
        assert np.array_equal(b0[c][h:n0[c]], b1[c][h:n1[c]]), c
        lag, err, n = synth.align_and_count_errors(b1[c][:n1[c]], txb[c], skip=h)
        assert err == 0 and n > 11000
        assert abs(st0[c].agc_gain - st1[c].agc_gain) < 1e-3 * st0[c].agc_gain and st0[c].offset == st1[c].offset


def test_folded_cody_waite_step_is_exact_for_every_phase(tmp_path):
    """The generator's fold_c12 option (gen_fll_asm.py; measured, not shipped: one slot less moved neither launch time) reduces the
    NCO phase with fma(-k, C1 + C2, x) instead of the oracle's fma(-k, C2, fma(-k, C1, x)).  C1 + C2 is a binary32 number, k = rint(x / pi) is -1, 0 or 1 for |x| <= pi and x - k*C1 is
    then exact, so the two are the same rounding of the same real number -- and here every binary32 value of [-pi, pi]
    (2 157 060 024 of them, both signs) is run through both forms."""
    import subprocess
    src = tmp_path / "fold.c"
    src.write_text(r'''
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
int main(void) {
    const float C1 = 3.140625f, C2 = 9.67502593994140625e-4f, C12 = 3.141592502593994140625f, pi = 3.1415926535f;
    uint32_t hi;
    memcpy(&hi, &pi, 4);
    long long bad = 0, n = 0;
    if ((double)C12 != (double)C1 + (double)C2) { puts("C1 + C2 is not representable"); return 1; }
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
    for (long long u = 0; u <= (long long)hi; u++) {
        for (int sgn = 0; sgn < 2; sgn++) {
            const uint32_t b = (uint32_t)u | (sgn ? 0x80000000u : 0u);
            float x;
            memcpy(&x, &b, 4);
            const float nk = -rintf(x * 0.318309886183790672f);
            const float two = fmaf(nk, C2, fmaf(nk, C1, x)), one = fmaf(nk, C12, x);
            n++;
            if (memcmp(&two, &one, 4)) bad++;
        }
    }
    printf("%lld %lld\n", n, bad);
    return 0;
}
''')
    exe = tmp_path / "fold"
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", str(src), "-o", str(exe), "-lm"], check=True)
    n, bad = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(n) == 2 * 1078530012 and int(bad) == 0


def test_rint_phase_wrap_is_exact_for_every_phase(tmp_path):
    """The generated FLL blocks wrap the loop phase as w = rint(x * WRAP_C), x = fma(-w, 2 pi, x) (three instructions) instead of
    PhaseControlLoop's `x > pi -> x - 2 pi, x < -pi -> x + 2 pi` (copysign, subtract, compare, select).  WRAP_C = 0x3e22f983 is
    the binary32 nearest 1 / (2 pi).  Every binary32 x of [-2 pi, 2 pi] (the loop's sums stay inside 1.5 pi) through both forms:
    the only difference is x = -0 -> +0, which a phase that starts at +0 can never reach (x + y is -0 only for two -0 operands)
    and which tetra_demod_set_state canonicalises."""
    import subprocess
    src = tmp_path / "wrap.c"
    src.write_text(r'''
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
int main(void) {
    const float pi = 3.1415926535f, twopi = pi - (-pi);
    float c; uint32_t cb = 0x3e22f983u; memcpy(&c, &cb, 4);
    float lim = 2.0f * pi; uint32_t hi; memcpy(&hi, &lim, 4);
    long long bad = 0, n = 0, negzero = 0;
    if (fabs((double)c - 1.0 / (2.0 * 3.14159265358979323846)) > 1e-8) return 1;
#pragma omp parallel for reduction(+ : bad, n, negzero) schedule(static)
    for (long long u = 0; u <= (long long)hi; u++) {
        for (int sgn = 0; sgn < 2; sgn++) {
            const uint32_t b = (uint32_t)u | (sgn ? 0x80000000u : 0u);
            float x; memcpy(&x, &b, 4);
            float ref = x;
            if (x > pi) ref = x - twopi; else if (x < -pi) ref = x + twopi;
            const float w = rintf(x * c);
            const float got = fmaf(-w, twopi, x);
            n++;
            if (memcmp(&ref, &got, 4)) { if (b == 0x80000000u) negzero++; else bad++; }
        }
    }
    printf("%lld %lld %lld\n", n, bad, negzero);
    return 0;
}
''')
    exe = tmp_path / "wrap"
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", str(src), "-o", str(exe), "-lm"], check=True)
    n, bad, negzero = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(n) > 2100000000 and int(bad) == 0 and int(negzero) == 1
