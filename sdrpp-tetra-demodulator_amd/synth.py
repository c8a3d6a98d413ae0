"""Synthetic TETRA pi/4-DQPSK IQ generator (numpy; test and bench input only).

Per channel: uniform random bits -> dibits -> differential phase steps with the
ETSI EN 300 392-2 map the reference's decoder side also uses
(src/decoder/src/phy/tetra_burst.c:99-117: 00 -> +pi/4, 01 -> +3pi/4,
11 -> -3pi/4, 10 -> -pi/4) -> root-raised-cosine (alpha 0.35) pulse shaping
evaluated directly at the receiver's 2 samples/symbol with a fractional timing
offset -> carrier offset, amplitude, AWGN.  Parameter ranges follow SURVEY.md
section 8(d): cfo U(-0.05, 0.05) rad/sample, timing U[0, 2) samples, amplitude
U(0.05, 1.0), Es/N0 25 dB.
"""
import numpy as np

PHASE_STEP = {0b00: 1, 0b01: 3, 0b11: -3, 0b10: -1}  # in units of pi/4
_STEP_LUT = np.array([PHASE_STEP[d] for d in range(4)], dtype=np.int64)

SPAN = 8  # pulse truncated to +-SPAN symbols


def rrc_pulse(t, beta):
    """Unit-energy root-raised-cosine impulse response, t in symbol periods."""
    t = np.asarray(t, dtype=np.float64)
    out = np.empty_like(t)
    eps = 1e-9
    z = np.abs(t) < eps
    s = np.abs(np.abs(t) - 1.0 / (4.0 * beta)) < eps
    r = ~(z | s)
    tr = t[r]
    out[r] = (np.sin(np.pi * tr * (1 - beta)) + 4 * beta * tr * np.cos(np.pi * tr * (1 + beta))) / (
        np.pi * tr * (1 - (4 * beta * tr) ** 2))
    out[z] = 1 - beta + 4 * beta / np.pi
    out[s] = (beta / np.sqrt(2)) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * beta)) +
                                   (1 - 2 / np.pi) * np.cos(np.pi / (4 * beta)))
    return out


def bits_to_symbols(bits):
    """bits uint8[2K] (MSB first per dibit) -> complex128[K] pi/4-DQPSK symbols (phase starts at 0)."""
    bits = np.asarray(bits, dtype=np.int64)
    d = (bits[0::2] << 1) | bits[1::2]
    steps = _STEP_LUT[d]
    ph = np.cumsum(steps) % 8
    return np.exp(1j * np.pi / 4 * ph)


def modulate(bits, n_samples, sps=2.0, beta=0.35, tau=0.0, ppm=0.0):
    """Shape the symbol stream of `bits` and sample it n_samples times at sps samples/symbol.

    Sample n sits at symbol time t_n = (n*(1+ppm*1e-6) + tau)/sps - SPAN, i.e. symbol 0's
    peak is SPAN symbols after the first sample's nominal origin so the stream starts cleanly.
    """
    syms = bits_to_symbols(bits)
    K = syms.shape[0]
    n = np.arange(n_samples, dtype=np.float64)
    t = (n * (1.0 + ppm * 1e-6) + tau) / sps - SPAN
    k0 = np.floor(t).astype(np.int64)
    out = np.zeros(n_samples, dtype=np.complex128)
    pad = np.concatenate([np.zeros(2 * SPAN + 2, complex), syms, np.zeros(2 * SPAN + 2, complex)])
    base = 2 * SPAN + 2
    if ppm == 0.0 and float(sps).is_integer():
        # polyphase: t - k takes only `sps` distinct fractional values
        isps = int(sps)
        frac = t - k0
        for j in range(-SPAN, SPAN + 1):
            pj = np.empty(n_samples)
            for ph in range(min(isps, n_samples)):          # (a stream shorter than one symbol has fewer phases than isps)
                pj[ph::isps] = rrc_pulse(frac[ph] - j, beta)
            idx = np.clip(k0 + j + base, 0, pad.shape[0] - 1)
            valid = (k0 + j >= 0) & (k0 + j < K)
            out += np.where(valid, pad[idx], 0) * pj
    else:
        for j in range(-SPAN, SPAN + 1):
            k = k0 + j
            valid = (k >= 0) & (k < K)
            idx = np.clip(k + base, 0, pad.shape[0] - 1)
            out += np.where(valid, pad[idx], 0) * rrc_pulse(t - k, beta)
    return out


def needed_bits(n_samples, sps=2.0):
    return 2 * (int(n_samples / sps) + 2 * SPAN + 4)


def gen_channel(n_samples, seed, sps=2.0, beta=0.35, cfo=None, tau=None, amp=None, esn0_db=25.0,
                ppm=0.0, phase0=None, bits=None):
    """One channel.  Returns (iq complex64[n_samples], tx_bits uint8[...], params dict)."""
    rng = np.random.default_rng(seed)
    cfo = rng.uniform(-0.05, 0.05) if cfo is None else cfo
    tau = rng.uniform(0.0, 2.0) if tau is None else tau
    amp = rng.uniform(0.05, 1.0) if amp is None else amp
    phase0 = rng.uniform(-np.pi, np.pi) if phase0 is None else phase0
    if bits is None:
        bits = rng.integers(0, 2, size=needed_bits(n_samples, sps), dtype=np.uint8)
    s = modulate(bits, n_samples, sps=sps, beta=beta, tau=tau, ppm=ppm)
    n = np.arange(n_samples, dtype=np.float64)
    s = s * amp * np.exp(1j * (cfo * n + phase0))
    if esn0_db is not None:
        sigma2 = amp * amp * sps / (10.0 ** (esn0_db / 10.0))
        s = s + np.sqrt(sigma2 / 2) * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
    return s.astype(np.complex64), np.asarray(bits, np.uint8), dict(cfo=cfo, tau=tau, amp=amp, phase0=phase0)


def gen_batch(n_channels, n_samples, base_seed=1234, **kw):
    """Channel-major batch: iq complex64[C][N], list of tx bit arrays, list of params."""
    iq = np.empty((n_channels, n_samples), np.complex64)
    txb, prm = [], []
    for c in range(n_channels):
        x, b, p = gen_channel(n_samples, base_seed + c, **kw)
        iq[c] = x
        txb.append(b)
        prm.append(p)
    return iq, txb, prm


def align_and_count_errors(rx_bits, tx_bits, max_lag=400, skip=0, window=None):
    """Find the lag L (rx[i] == tx[i-L]) that minimises errors over the compared stretch.

    Compares rx_bits[skip:] only; returns (best_lag, n_errors, n_compared).
    """
    rx = np.asarray(rx_bits, np.uint8)
    tx = np.asarray(tx_bits, np.uint8)
    best = (None, 1 << 60, 0)
    for lag in range(-max_lag, max_lag + 1):
        lo = max(skip, lag, 0)
        hi = min(rx.shape[0], tx.shape[0] + lag)
        if window is not None:
            hi = min(hi, lo + window)
        if hi - lo < 64:
            continue
        err = int(np.count_nonzero(rx[lo:hi] != tx[lo - lag:hi - lag]))
        if err < best[1]:
            best = (lag, err, hi - lo)
    return best


# EN 300 392-2 9.4.4.3.2 / 9.4.4.3.4: normal training sequences 1, 2 and the synchronisation training sequence
TRAIN_NORM_1 = np.array([1,1, 0,1, 0,0, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,1, 0,0], np.uint8)
TRAIN_NORM_2 = np.array([0,1, 1,1, 1,0, 1,0, 0,1, 0,0, 0,0, 1,1, 0,1, 1,1, 1,0], np.uint8)
TRAIN_SYNC = np.array([1,1, 0,0, 0,0, 0,1, 1,0, 0,1, 1,1, 0,0, 1,1, 1,0, 1,0, 0,1, 1,1, 0,0, 0,0, 0,1, 1,0, 0,1, 1,1], np.uint8)


def gen_slot_bits(n_slots, seed):
    """A continuous downlink as far as the burst synchroniser is concerned: 510-bit slots of random payload with the
    synchronisation training sequence at bit 214 in every 4th slot and normal training sequence 1 (2 in odd slots) at bit
    244 in the others (EN 300 392-2 9.4.4.2.5/6).  Payload blocks are random, i.e. not channel coded: bench/profiling
    input for the demodulator -> synchroniser -> demultiplexer -> decoder chain, where only the work matters."""
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, (n_slots, 510), dtype=np.uint8)
    for s in range(n_slots):
        if s % 4 == 0:
            bits[s, 214:214 + 38] = TRAIN_SYNC
        else:
            bits[s, 244:244 + 22] = TRAIN_NORM_2 if s % 2 else TRAIN_NORM_1
    return bits.reshape(-1)
