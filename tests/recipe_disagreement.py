"""Where "bit-exact" ends: the two float recipes of the CPU oracle against each other, on the plugin's own parameters, versus Es/N0.

TEST INFRASTRUCTURE (imports oracle/): tests/test_recipe_disagreement.py asserts on it, profiles/recipe_disagreement.py runs the full
sweep and writes profiles/r05/recipe_disagreement.{json,md}.

The HIP kernels equal the oracle's CONTRACT recipe bit for bit (fmaf chains, one polynomial sincos, conjugate-pair band-edge sums);
the reference's compiled objects equal its REFERENCE_FLOATS recipe bit for bit (libm phasors, plain `acc += a * b`, two separate
complex band-edge dots: /root/reference src/dsp/fll.cpp:135-149, complex_fd.cpp:98-145 -- an x86 build without VOLK SIMD kernels;
a VOLK build differs from THAT the same way).  Both are the same algorithm in binary32 with another rounding sequence, so their
decision-directed loops (Costas, timing, FLL) see symbols that differ in the last bits -- and a decision that sits on the boundary
falls one way in one recipe and the other way in the other.  During acquisition that can even send the loops to another of their
equivalent lock points for a while; after lock it needs noise that puts a symbol within ~1e-3 of a decision boundary, i.e. it
happens at the rate of true bit errors x a small factor, and the two recipes then make DIFFERENT but EQUALLY MANY errors.

Per Es/N0: C channels of the BASELINE generator (random carrier offset, timing, level; src/main.cpp:35-44,78-84 parameters), two
seconds each; "after lock" = the second second, of the channels that HAVE locked by then: a channel whose true bit errors in that
second exceed ten times the median channel's + 50 in either recipe is still acquiring (about one in a thousand at 20 dB: low level and
large offset drawn together; what the two recipes put out there is garbage against garbage) and is counted apart.  Reported: channels / bits that differ before and after lock, and the true bit errors (against the transmitted bits) of each
recipe after lock.
"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SAMPLES = 72000          # two seconds at 36 ksps
ESN0_DB = (30.0, 25.0, 20.0, 15.0, 12.0, 10.0, 8.0)


def _one_channel(args):
    esn0_db, seed = args
    import tetra_amd
    synth = tetra_amd.pkg.synth
    from oracle import binding as ob
    iq, txb, _ = synth.gen_channel(N_SAMPLES, seed, esn0_db=esn0_db)
    c = ob.Oracle().process(iq)["bits"]
    r = ob.Oracle(reference_floats=True).process(iq)["bits"]
    n = min(c.size, r.size)
    half = n // 2
    d = c[:n] != r[:n]
    out = dict(seed=seed, n_bits=int(n), count_differs=int(c.size != r.size), diff_before=int(d[:half].sum()), diff_after=int(d[half:].sum()))
    for name, bits in (("contract", c), ("ref_float", r)):
        # true bit errors in the second second: align there (the lag is the chain's constant delay; a timing slip would show as errors)
        lag, err, cmp_n = synth.align_and_count_errors(bits, txb, skip=bits.size // 2)
        out["err_" + name] = int(err)
        out["cmp_" + name] = int(cmp_n)
    return out


def sweep(esn0_list=ESN0_DB, channels=64, base_seed=52000, workers=None):
    """Returns {esn0_db: summary dict}."""
    workers = workers or min(8, os.cpu_count() or 1)
    jobs = [(e, base_seed + 1000 * i + c) for i, e in enumerate(esn0_list) for c in range(channels)]
    with ProcessPoolExecutor(max_workers=workers) as ex:
        res = list(ex.map(_one_channel, jobs, chunksize=4))
    out = {}
    for i, e in enumerate(esn0_list):
        rows = res[i * channels:(i + 1) * channels]
        limit = {k: 10.0 * float(np.median([r["err_" + k] for r in rows])) + 50.0 for k in ("contract", "ref_float")}
        lk = [r for r in rows if r["err_contract"] <= limit["contract"] and r["err_ref_float"] <= limit["ref_float"]]
        out[e] = dict(
            esn0_db=e, channels=channels, seconds=N_SAMPLES / 36000.0,
            channels_locked=len(lk),
            channels_differing_before_lock=sum(r["diff_before"] > 0 for r in rows),
            bits_differing_before_lock=sum(r["diff_before"] for r in rows),
            locked_channels_differing_after_lock=sum(r["diff_after"] > 0 for r in lk),
            locked_bits_differing_after_lock=sum(r["diff_after"] for r in lk),
            unlocked_bits_differing_in_second_second=sum(r["diff_after"] for r in rows) - sum(r["diff_after"] for r in lk),
            channels_with_other_symbol_count=sum(r["count_differs"] for r in rows),
            locked_true_errors_contract=sum(r["err_contract"] for r in lk), locked_true_errors_ref_float=sum(r["err_ref_float"] for r in lk),
            locked_bits_compared=sum(r["cmp_contract"] for r in lk),
        )
    return out


def ber_gap_in_sigmas(row):
    """|BER_contract - BER_ref_float| over the LOCKED channels in units of the binomial standard error of their difference
    (independent-errors bound)."""
    n = max(row["locked_bits_compared"], 1)
    pc, pr = row["locked_true_errors_contract"] / n, row["locked_true_errors_ref_float"] / n
    p = 0.5 * (pc + pr)
    se = np.sqrt(max(2.0 * p * (1.0 - p) / n, 1e-300))
    return abs(pc - pr) / se if p > 0 else 0.0


def markdown(res):
    lines = ["| Es/N0 dB | channels (locked in both) | differ before lock (ch / bits) | differ after lock, locked channels (ch / bits) | true errors after lock, locked channels: contract | ref-float | BER contract | BER ref-float | gap in σ |",
             "|---|---|---|---|---|---|---|---|---|"]
    for e in sorted(res, key=float, reverse=True):
        r = res[e]
        n = max(r["locked_bits_compared"], 1)
        lines.append("| %g | %d (%d) | %d / %d | %d / %d | %d | %d | %.2e | %.2e | %.2f |" % (
            float(e), r["channels"], r["channels_locked"], r["channels_differing_before_lock"], r["bits_differing_before_lock"],
            r["locked_channels_differing_after_lock"], r["locked_bits_differing_after_lock"], r["locked_true_errors_contract"],
            r["locked_true_errors_ref_float"], r["locked_true_errors_contract"] / n, r["locked_true_errors_ref_float"] / n, ber_gap_in_sigmas(r)))
    return "\n".join(lines)
