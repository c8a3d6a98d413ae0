"""Generates tests/golden/refshim_vectors.npz: inputs and the outputs of the REFERENCE'S OWN src/dsp code for them.

Container only (needs /root/reference): the reference's pi4dqpsk.cpp / fll.cpp / complex_fd.cpp / pi4dqpsk_costas.cpp /
dqpsk_sym_extr.cpp / bit_unpacker.cpp are compiled where they lie against the stand-in SDR++ core headers under
tests/refshim/ (see tests/test_reference_shim.py) and run on synthetic IQ.  The fixture is DATA (inputs + expected outputs);
it lets the oracle and the GPU be compared with what the reference's code computed on machines that do not have the
reference (the GPU box).  Because the core headers are stand-ins, this is evidence, not a parity pin.
Run from the repo root:  python tests/golden/make_refshim_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tetra_amd  # noqa: E402
from oracle import binding as ob  # noqa: E402
from tests import test_reference_shim as T  # noqa: E402


def state_vec(r):
    """Loop state of the reference's objects after the last call, as float64 [11] in the order of T.STATE_FLOATS + T.STATE_INTS
    (a float32 value is exactly representable; offset / prev are small integers)."""
    st = r.state()
    return np.array([float(st[k]) for k in T.STATE_FLOATS + T.STATE_INTS], np.float64)


def main():
    synth = tetra_amd.pkg.synth
    L = T.load_reference_build()
    out = {}
    out["state_fields"] = np.array(T.STATE_FLOATS + T.STATE_INTS)
    # 1. the survey's probe scenario
    # 2. a channel of the BASELINE generator, one second, processed in 180-sample calls like SDR++ delivers them
    # 3. reset() and setters mid-stream (the reference's own reset / setter code)
    iq1, _, _ = synth.gen_channel(40060, 7, cfo=0.03, tau=5 / 16.0, amp=0.2, esn0_db=25.0)
    r = T.RefChain(L, ob.default_cfg())
    out["probe_iq"] = iq1
    out["probe_sym"], out["probe_bits"] = r.process(iq1)
    out["probe_state"] = state_vec(r)
    r.close()
    iq2, _, _ = synth.gen_channel(12000, 1234)
    r = T.RefChain(L, ob.default_cfg())
    parts = [r.process(iq2[i:i + 180]) for i in range(0, len(iq2), 180)]
    out["chunked_state"] = state_vec(r)
    r.close()
    out["chunked_iq"] = iq2
    out["chunked_sym"] = np.concatenate([p[0] for p in parts])
    out["chunked_bits"] = np.concatenate([p[1] for p in parts])
    iq3, _, _ = synth.gen_channel(24000, 77, cfo=0.01, tau=0.4, amp=0.3)
    r = T.RefChain(L, ob.default_cfg())
    a = r.process(iq3[:9001])
    out["ctl_state0"] = state_vec(r)
    r.reset()
    b = r.process(iq3[9001:16000])
    out["ctl_state1"] = state_vec(r)
    for pid, v in ((4, 0.03), (5, 0.004), (6, 0.008), (7, 2e-4), (8, 0.02), (9, 0.02), (2, 49)):
        r.set_param(pid, v)
    c = r.process(iq3[16000:])
    out["ctl_state2"] = state_vec(r)
    r.close()
    out["ctl_iq"] = iq3
    out["ctl_cuts"] = np.array([0, 9001, 16000, 24000], np.int32)
    out["ctl_setters"] = np.array([(4, 0.03), (5, 0.004), (6, 0.008), (7, 2e-4), (8, 0.02), (9, 0.02), (2, 49)], np.float64)
    for k, (sym, bits) in enumerate((a, b, c)):
        out["ctl_sym%d" % k] = sym
        out["ctl_bits%d" % k] = bits
    # 4. BASELINE config 5's rate: the chain created at 50 ksps (2.78 samples per symbol), one call
    cfg50 = ob.default_cfg()
    cfg50.samplerate = 50000.0
    iq4, _, _ = synth.gen_channel(14000, 321, sps=50000.0 / 18000.0, cfo=0.015)
    r = T.RefChain(L, cfg50)
    out["rate50_iq"] = iq4
    out["rate50_sym"], out["rate50_bits"] = r.process(iq4)
    out["rate50_state"] = state_vec(r)
    r.close()
    # 5. eight chains side by side (one reference object set per channel), for the batched kernel: [8][7000] in, ragged out
    iq5 = np.stack([synth.gen_channel(7000, 900 + c)[0] for c in range(8)])
    syms, bitss, states = [], [], []
    for c in range(8):
        r = T.RefChain(L, ob.default_cfg())
        s_, b_ = r.process(iq5[c])
        states.append(state_vec(r))
        r.close()
        syms.append(s_)
        bitss.append(b_)
    out["multi8_iq"] = iq5
    out["multi8_nsym"] = np.array([len(s_) for s_ in syms], np.int32)
    out["multi8_sym"] = np.concatenate(syms)
    out["multi8_bits"] = np.concatenate(bitss)
    out["multi8_state"] = np.stack(states)
    # 6. setRRCParams(49, 0.35) mid-stream: the roll-off stays a double (only setRRCBeta(int) truncates)
    iq6, _, _ = synth.gen_channel(12000, 79, cfo=0.012, tau=0.9, amp=0.4)
    r = T.RefChain(L, ob.default_cfg())
    a6 = r.process(iq6[:6000])
    r.set_rrc_params(49, 0.35)
    b6 = r.process(iq6[6000:])
    out["rrcp_state"] = state_vec(r)
    r.close()
    out["rrcp_iq"] = iq6
    out["rrcp_sym0"], out["rrcp_bits0"] = a6
    out["rrcp_sym1"], out["rrcp_bits1"] = b6
    # 7. random parameter sets (rates 1.8 ... 4 samples per symbol at create, tap counts 2 ... 72, roll-off, every loop constant;
    #    the draw of profiles/fuzz_parity.py): eight sets on which the reference code and the oracle make the same decisions
    #    throughout -- on ~3 % of random draws their float recipes (libm sine and plain sums / polynomial and fmaf chains) part
    #    ways at a boundary decision, profiles/fuzz_refshim_cpu.py; those are skipped here, and the skip is counted
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import fuzz_parity as F  # noqa: E402
    rng = np.random.default_rng(20260927)
    k = skipped = 0
    while k < 8:
        sps, prm = F.draw_params(rng)
        cfg = F.oracle_cfg(prm)
        seed = int(rng.integers(0, 1 << 30))
        if sps < 1.8 or sps * (1 - cfg.omega_rel_limit) - abs(cfg.mu_gain) < 1.0:
            continue
        iq7, _, _ = synth.gen_channel(6000, seed, sps=sps)
        r = T.RefChain(L, cfg)
        sym7, bits7 = r.process(iq7)
        st7 = state_vec(r)
        r.close()
        o = ob.Oracle(cfg).process(iq7)
        if len(o["bits"]) != len(bits7) or not np.array_equal(o["bits"], bits7):
            skipped += 1
            continue
        out["rand%d_cfg" % k] = np.array([cfg.symbolrate, cfg.samplerate, cfg.rrc_tap_count, cfg.rrc_beta, cfg.agc_rate,
                                          cfg.costas_bandwidth, cfg.fll_bandwidth, cfg.omega_gain, cfg.mu_gain, cfg.omega_rel_limit], np.float64)
        out["rand%d_iq" % k], out["rand%d_sym" % k], out["rand%d_bits" % k] = iq7, sym7, bits7
        out["rand%d_state" % k] = st7
        k += 1
    out["rand_skipped"] = np.array([skipped], np.int32)
    # 8. below one sample per symbol step (COMPLEX_FD emits several symbols from one offset, complex_fd.cpp:98-145: floor(mu) = 0):
    #    the chain created at 1.0 and at 0.9 samples per symbol; three ragged calls each
    for tag, rate, seed in (("sps100", 18000.0, 555), ("sps090", 16200.0, 557)):
        cfg = ob.default_cfg()
        cfg.samplerate = rate
        iq8, _, _ = synth.gen_channel(3000, seed, sps=1.02)
        r = T.RefChain(L, cfg)
        parts = [r.process(iq8[a:b]) for a, b in ((0, 1), (1, 1200), (1200, 3000))]
        out[tag + "_iq"] = iq8
        out[tag + "_rate"] = np.array([rate])
        out[tag + "_sym"] = np.concatenate([p_[0] for p_ in parts])
        out[tag + "_bits"] = np.concatenate([p_[1] for p_ in parts])
        out[tag + "_state"] = state_vec(r)
        r.close()
    # 9. filters beyond 72 taps (PI4DQPSK::init takes any count, pi4dqpsk.cpp:11-30): 101 and 129 taps, three ragged calls each; the
    #    129-tap one at 1.0 samples per symbol (long filter AND several symbols from one offset).  Seeds on which the contract-mode
    #    oracle makes the reference code's decisions throughout (see 7.)
    for tag, taps, rate, sps in (("long101", 101, 36000.0, 2.0), ("long129", 129, 36000.0, 2.0), ("long129s", 129, 18000.0, 1.02)):
        cfg = ob.default_cfg()
        cfg.rrc_tap_count = taps
        cfg.samplerate = rate
        seed = 700
        while True:
            iq9, _, _ = synth.gen_channel(5000, seed, sps=sps, cfo=0.01)
            r = T.RefChain(L, cfg)
            parts = [r.process(iq9[a_:b_]) for a_, b_ in ((0, 7), (7, 1900), (1900, 5000))]
            st9 = state_vec(r)
            r.close()
            bits9 = np.concatenate([p_[1] for p_ in parts])
            o = ob.Oracle(cfg).process(iq9)
            if len(o["bits"]) == len(bits9) and np.array_equal(o["bits"], bits9):
                break
            seed += 1
        out[tag + "_iq"] = iq9
        out[tag + "_cfg"] = np.array([taps, rate], np.float64)
        out[tag + "_sym"] = np.concatenate([p_[0] for p_ in parts])
        out[tag + "_bits"] = bits9
        out[tag + "_state"] = st9
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim_vectors.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
