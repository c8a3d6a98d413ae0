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


# ---------------------------------------------------------------------------------------------------------------------
# Transmit side of the lower MAC (EN 300 392-2 section 8.2) and the two continuous downlink bursts (9.4.4.2.5 / .6): CODED
# downlink streams for the receive chain's known-answer checks (tests, bench.py) -- every block the chain decodes carries a CRC
# that must come out good, and its type-1 bits are known.  numpy, vectorised over blocks.  The reference has these steps as
# C primitives (src/decoder/src/lower_mac/crc_simple.c:62-103, tetra_conv_enc.c:45-95 mother code, :99-137 + :204-226 rate-2/3
# puncturer, tetra_interleave.c:36-48, tetra_scramb.c:34-85,87-99; burst layouts phy/tetra_burst.c:171-269); tests/test_synth_tx.py
# holds this generator bit for bit against those primitives compiled from the reference (oracle/_ref).
# ---------------------------------------------------------------------------------------------------------------------
# (type345 bits, type2 bits, type1 bits, interleaver a) per block kind -- tetra_blk_param[], lower_mac/tetra_lower_mac.c:58-105
TX_BLK = {"sb1": (120, 80, 60, 11), "sb2": (216, 144, 124, 101), "ndb": (216, 144, 124, 101), "schf": (432, 288, 268, 103)}
SCRAMB_INIT_SB1 = 3
_Q_BITS = np.array([1,0, 1,1, 0,1, 1,1, 0,0, 0,0, 0,1, 1,0, 1,0, 1,1, 0,1], np.uint8)          # 9.4.4.3.2 normal training sequence 3
_F_BITS = np.array([1] * 8 + [0] * 64 + [1] * 8, np.uint8)                                     # 9.4.4.3.1 frequency correction field


def tx_scramb_code(mcc, mnc, colour):
    """Scrambling code of a cell (EN 300 392-2 8.2.5.2; tetra_scramb.c:87-99)."""
    return ((((colour & 0x3f) | ((mnc & 0x3fff) << 6) | ((mcc & 0x3ff) << 20)) << 2) | SCRAMB_INIT_SB1) & 0xffffffff


def tx_scramb_seq(codes, n):
    """codes uint32 [m] -> the first n scrambling bits per code, uint8 [m][n] (Fibonacci LFSR, taps 32 26 23 22 16 12 11 10 8 7 5 4 2 1)."""
    lfsr = np.asarray(codes, np.uint64).copy() & np.uint64(0xffffffff)
    out = np.zeros((lfsr.shape[0], n), np.uint8)
    taps = (32, 26, 23, 22, 16, 12, 11, 10, 8, 7, 5, 4, 2, 1)
    for i in range(n):
        bit = np.zeros_like(lfsr)
        for t in taps:
            bit ^= lfsr >> np.uint64(32 - t)
        bit &= np.uint64(1)
        lfsr = (lfsr >> np.uint64(1)) | (bit << np.uint64(31))
        out[:, i] = bit.astype(np.uint8)
    return out


def tx_crc16(bits):
    """bits uint8 [m][n] -> the 16 CRC bits appended to a type-1 block, uint8 [m][16]: ones' complement of the CCITT CRC
    (x^16 + x^12 + x^5 + 1, preset 0xffff), MSB first."""
    bits = np.asarray(bits, np.uint8)
    crc = np.full(bits.shape[0], 0xffff, np.uint32)
    for i in range(bits.shape[1]):
        crc ^= bits[:, i].astype(np.uint32) << 15
        top = (crc & 0x8000) != 0
        crc = (crc << 1) & 0xffff
        crc[top] ^= 0x1021
    crc = (~crc) & 0xffff
    return ((crc[:, None] >> (15 - np.arange(16))[None, :]) & 1).astype(np.uint8)


def tx_encode(kind, type1, codes):
    """type-1 bits uint8 [m][type1 bits] + scrambling codes uint32 [m] (ignored for "sb1": always 3) -> type-5 bits uint8 [m][type345 bits]:
    CRC16 + 4 tail bits, rate-1/4 mother code (G1 = 1+D+D4, G2 = 1+D2+D3+D4, G3 = 1+D+D2+D4, G4 = 1+D+D3+D4), rate-2/3 puncturing,
    block interleaving, scrambling."""
    n345, n2, n1, a = TX_BLK[kind]
    t1 = np.asarray(type1, np.uint8).reshape(-1, n1)
    m = t1.shape[0]
    t2 = np.zeros((m, n2 + 4), np.uint8)                  # 4 leading zeros = the encoder's initial state
    t2[:, 4:4 + n1] = t1
    t2[:, 4 + n1:4 + n1 + 16] = tx_crc16(t1)              # then 4 zero tail bits
    b, d1, d2, d3, d4 = t2[:, 4:], t2[:, 3:-1], t2[:, 2:-2], t2[:, 1:-3], t2[:, :-4]
    mother = np.empty((m, 4 * n2), np.uint8)
    mother[:, 0::4] = b ^ d1 ^ d4
    mother[:, 1::4] = b ^ d2 ^ d3 ^ d4
    mother[:, 2::4] = b ^ d1 ^ d2 ^ d4
    mother[:, 3::4] = b ^ d1 ^ d3 ^ d4
    j = np.arange(1, n345 + 1)
    P = np.array([0, 1, 2, 5])
    k = 8 * ((j - 1) // 3) + P[j - 3 * ((j - 1) // 3)]
    t3 = mother[:, k - 1]
    i = np.arange(1, n345 + 1)
    t4 = np.empty_like(t3)
    t4[:, (a * i) % n345] = t3                            # out[k - 1] = in[i - 1], k = 1 + (a i mod K)
    c = np.full(m, SCRAMB_INIT_SB1, np.uint32) if kind == "sb1" else np.asarray(codes, np.uint32).reshape(m)
    uniq, inv = np.unique(c, return_inverse=True)
    return t4 ^ tx_scramb_seq(uniq, n345)[inv]


_PHASE_OF_DIBIT = np.array([1, 3, -1, -3])               # dibit value (first bit << 1 | second) -> phase step in pi/4 (EN 300 392-2 5.5.2.3 = PHASE_STEP)


def _phase_adj(burst, n1, n2):
    """9.4.4.3.6: the two bits whose symbol brings the phase accumulated over symbols n1 .. n2 (1-based) back to a multiple of 2 pi."""
    seg = burst[2 * (n1 - 1): 2 * n2].astype(np.int64)
    total = int(_PHASE_OF_DIBIT[(seg[0::2] << 1) | seg[1::2]].sum())
    adj = (-total) % 8
    adj = adj - 8 if adj > 3 else adj                     # in {-3, -1, 1, 3} (an odd number of symbols)
    return {1: (0, 0), 3: (0, 1), -1: (1, 0), -3: (1, 1)}[adj]


def tx_sync_burst(sb, bb, bkn2):
    """9.4.4.2.6 synchronisation continuous downlink burst (phy/tetra_burst.c:171-219): 510 bits."""
    b = np.concatenate([_Q_BITS[10:], [0, 0], _F_BITS, sb, TRAIN_SYNC, bb, bkn2, [0, 0], _Q_BITS[:10]]).astype(np.uint8)
    b[12:14] = _phase_adj(b, 8, 108)
    b[498:500] = _phase_adj(b, 109, 249)
    return b


def tx_norm_burst(bkn1, bb, bkn2, two_log_chan):
    """9.4.4.2.5 normal continuous downlink burst (phy/tetra_burst.c:222-269): 510 bits."""
    train = TRAIN_NORM_2 if two_log_chan else TRAIN_NORM_1
    b = np.concatenate([_Q_BITS[10:], [0, 0], bkn1, bb[:14], train, bb[14:], bkn2, [0, 0], _Q_BITS[:10]]).astype(np.uint8)
    b[12:14] = _phase_adj(b, 8, 122)
    b[498:500] = _phase_adj(b, 123, 249)
    return b


def tdma_time_of_slot(s):
    """(tn, fn, mn) of slot s of a downlink that starts at timeslot 1 of frame 1 of multiframe 1 (4 slots per frame, 18 frames per
    multiframe, 60 multiframes per hyperframe)."""
    return s % 4 + 1, (s // 4) % 18 + 1, (s // 72) % 60 + 1


def gen_downlink(n_slots, seed, cell=(262, 1, 5), slot0=0):
    """A coded continuous downlink of one cell: slot s (absolute s + slot0) carries
         s % 4 == 0   SYNC burst: SB1 = SYNC PDU (60 type-1 bits: the cell's colour code at bits 4..9, the slot's TN-1 / FN / MN at
                      10..11 / 12..16 / 17..22, MCC at 31..40, MNC at 41..54 -- the fields tetra_lower_mac.c:246-275 reads --, the rest
                      random), AACH (30 random bits, scrambled), SB2 (124 random type-1 bits)
         s % 4 == 2   normal burst with two logical channels: NDB block 1 + block 2 (124 random type-1 bits each), AACH
         otherwise    normal burst with one logical channel: SCH/F (268 random type-1 bits), AACH
    everything but SB1 scrambled with the cell's code.  Returns (bits uint8 [n_slots * 510], sent) with sent = {kind: [(slot,
    type-1 bits)]} for kind in sb1 / sb2 / ndb1 / ndb2 / schf / bbk (bbk: the 30 descrambled AACH bits)."""
    rng = np.random.default_rng(seed)
    mcc, mnc, cc = cell
    code = tx_scramb_code(mcc, mnc, cc)
    slots = np.arange(n_slots)
    s_sync, s_ndb = slots[slots % 4 == 0], slots[slots % 4 == 2]
    s_schf = slots[(slots % 4 == 1) | (slots % 4 == 3)]
    def field(v, n):
        return [(v >> (n - 1 - i)) & 1 for i in range(n)]
    pdu = rng.integers(0, 2, (len(s_sync), 60), dtype=np.uint8)
    for r, s in enumerate(s_sync):
        tn, fn, mn = tdma_time_of_slot(int(s) + slot0)
        pdu[r, 4:10], pdu[r, 10:12], pdu[r, 12:17], pdu[r, 17:23] = field(cc, 6), field(tn - 1, 2), field(fn, 5), field(mn, 6)
        pdu[r, 31:41], pdu[r, 41:55] = field(mcc, 10), field(mnc, 14)
    t1 = {"sb1": pdu, "sb2": rng.integers(0, 2, (len(s_sync), 124), dtype=np.uint8),
          "ndb1": rng.integers(0, 2, (len(s_ndb), 124), dtype=np.uint8), "ndb2": rng.integers(0, 2, (len(s_ndb), 124), dtype=np.uint8),
          "schf": rng.integers(0, 2, (len(s_schf), 268), dtype=np.uint8), "bbk": rng.integers(0, 2, (n_slots, 30), dtype=np.uint8)}
    t5 = {k: tx_encode({"ndb1": "ndb", "ndb2": "ndb"}.get(k, k), v, np.full(len(v), code, np.uint32)) for k, v in t1.items() if k != "bbk"}
    bb = t1["bbk"] ^ tx_scramb_seq([code], 30)[0][None, :]
    out = np.zeros((n_slots, 510), np.uint8)
    for r, s in enumerate(s_sync):
        out[s] = tx_sync_burst(t5["sb1"][r], bb[s], t5["sb2"][r])
    for r, s in enumerate(s_ndb):
        out[s] = tx_norm_burst(t5["ndb1"][r], bb[s], t5["ndb2"][r], 1)
    for r, s in enumerate(s_schf):
        out[s] = tx_norm_burst(t5["schf"][r][:216], bb[s], t5["schf"][r][216:], 0)
    sent = {"sb1": list(zip(s_sync.tolist(), t1["sb1"])), "sb2": list(zip(s_sync.tolist(), t1["sb2"])),
            "ndb1": list(zip(s_ndb.tolist(), t1["ndb1"])), "ndb2": list(zip(s_ndb.tolist(), t1["ndb2"])),
            "schf": list(zip(s_schf.tolist(), t1["schf"])), "bbk": list(zip(slots.tolist(), t1["bbk"]))}
    return out.reshape(-1), sent


# ---------------------------------------------------------------------------------------------------------------------
# Counter-based per-channel streams: channel c of a bank is fully determined by its own seed (base seed + global channel index) --
# bits and channel parameters are a hash of (seed, position), so any channel can be regenerated on its own, on the CPU (here) or
# for a whole bank on the GPU (synth_gpu.py, the same integer arithmetic in torch), in any order and any chunking.
# ---------------------------------------------------------------------------------------------------------------------
def hash_u32(seed, k):
    """32-bit mix of (seed, k); seed a non-negative int < 2^31, k int64 array (or int) >= 0.  All intermediate values < 2^63."""
    x = (np.asarray(seed, np.int64) * 2654435761 + np.asarray(k, np.int64) * 40503 + 12345) & 0xffffffff
    for _ in range(3):
        x ^= x >> 16
        x = (x * 0x45d9f3b) & 0xffffffff
    x ^= x >> 16
    return x


def hash_bits(seed, n):
    """The n transmitted bits of the channel with this seed."""
    return ((hash_u32(seed, np.arange(n, dtype=np.int64)) >> 7) & 1).astype(np.uint8)


HASH_PARAM_BASE = 1 << 40          # positions of the channel parameters in the hash stream (far beyond any bit position)


def hash_params(seed):
    """Channel parameters of the channel with this seed, SURVEY.md 8(d) ranges: cfo U(-0.05, 0.05) rad/sample, tau U[0, 2) samples,
    amp U(0.05, 1.0), phase0 U(-pi, pi)."""
    u = hash_u32(seed, HASH_PARAM_BASE + np.arange(4, dtype=np.int64)).astype(np.float64) / 4294967296.0
    return dict(cfo=-0.05 + 0.1 * u[0], tau=2.0 * u[1], amp=0.05 + 0.95 * u[2], phase0=-np.pi + 2.0 * np.pi * u[3])
