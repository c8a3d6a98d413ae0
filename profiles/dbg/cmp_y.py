"""Debug aid: GPU vs oracle on one small batch; prints where RRC output / bits / state differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tetra_amd
pkg = tetra_amd.pkg
if os.environ.get("TETRA_LIB"):
    pkg.build.LIB = os.path.abspath(os.environ["TETRA_LIB"])
    pkg.build.is_stale = lambda: False
from oracle import binding as ob
Cn, N = int(sys.argv[1]), int(sys.argv[2])
iq, _, _ = pkg.synth.gen_batch(Cn, N, base_seed=1000 + Cn)
d = pkg.Demodulator(Cn, N, flags=2)
bits, nb, sym = d.process(iq, want_sym=True)
y = d.read_rrc_out(N)
for c in range(min(Cn, 4)):
    o = ob.Oracle()
    r = o.process(iq[c], stages=True)
    yd = np.flatnonzero(y[c].view(np.uint32).reshape(-1, 2).any(axis=1) != 0)
    diff = np.flatnonzero((y[c].view(np.uint32).reshape(-1, 2) != r["y"].view(np.uint32).reshape(-1, 2)).any(axis=1))
    print("ch", c, "y diffs:", diff.size, "first", diff[:12], "last", diff[-5:] if diff.size else None,
          "max abs", float(np.abs(y[c] - r["y"]).max()), "rel", float(np.abs(y[c] - r["y"]).max() / np.abs(r["y"]).max()))
    n = min(nb[c], r["bits"].size)
    bd = np.flatnonzero(bits[c][:n] != r["bits"][:n])
    print("   n_bits", nb[c], r["bits"].size, "bit diffs", bd.size, bd[:10])
    st = d.get_state(c)
    print("   state fll", st.fll_phase, o.st.fll_phase, st.fll_freq, o.st.fll_freq, "agc", st.agc_gain, o.st.agc_gain)
    hg = np.array(st.hist[:], np.float32)[32:]; ho = np.array(o.st.hist[:128], np.float32)
    hd = np.flatnonzero(hg.view(np.uint32) != ho.view(np.uint32))
    print("   hist diffs", hd.size, hd[:10])
    if hd.size:
        xi = sorted(set((hd // 2 + 16 - 80 + N).tolist()))
        print("   first differing x index:", xi[:12])
    if hd.size and os.environ.get("VERBOSE"):
        hgc = hg.view(np.complex64); hoc = ho.view(np.complex64)
        for k in range(hgc.size):
            xi = k + 16 - 80 + N
            if xi >= 0 and hgc[k] != hoc[k]:
                print("      x[%d] gpu %r ref %r  diff %.3e" % (xi, hgc[k], hoc[k], abs(hgc[k] - hoc[k])))
