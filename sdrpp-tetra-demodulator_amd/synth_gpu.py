"""Synthetic TETRA pi/4-DQPSK IQ for a whole bank, generated ON the GPU with torch (test and bench input only; torch is plumbing
here, never the measured path).  The same signal model as synth.py -- synth.modulate's pulse-shaping sum at 2 samples per symbol in
float64, then carrier offset, amplitude, phase and AWGN per channel -- for thousands of channels at once, each from its OWN seed:
bits and channel parameters are synth.hash_u32 of (seed, position), evaluated here with the same integer arithmetic, so the CPU can
regenerate any single channel's transmitted bits (synth.hash_bits) for a known-answer check."""
import math

import numpy as np

from . import synth


def _hash_u32(torch, seed, k):
    """synth.hash_u32 on int64 tensors (every intermediate value < 2^63: no overflow, no unsigned types needed)."""
    x = (seed * 2654435761 + k * 40503 + 12345) & 0xffffffff
    for _ in range(3):
        x = x ^ (x >> 16)
        x = (x * 0x45d9f3b) & 0xffffffff
    return x ^ (x >> 16)


def hash_bits(torch, device, seeds, n):
    """seeds int64 [C] -> uint8 [C][n], row c = synth.hash_bits(seeds[c], n)."""
    seeds = torch.as_tensor(seeds, dtype=torch.int64, device=device)
    k = torch.arange(n, dtype=torch.int64, device=device)
    return ((_hash_u32(torch, seeds[:, None], k[None, :]) >> 7) & 1).to(torch.uint8)


def hash_params(torch, device, seeds):
    """seeds [C] -> dict of float64 tensors [C]: cfo, tau, amp, phase0 = synth.hash_params per channel."""
    seeds = torch.as_tensor(seeds, dtype=torch.int64, device=device)
    k = synth.HASH_PARAM_BASE + torch.arange(4, dtype=torch.int64, device=device)
    u = _hash_u32(torch, seeds[:, None], k[None, :]).to(torch.float64) / 4294967296.0
    return dict(cfo=-0.05 + 0.1 * u[:, 0], tau=2.0 * u[:, 1], amp=0.05 + 0.95 * u[:, 2], phase0=-math.pi + 2.0 * math.pi * u[:, 3])


def _rrc(torch, t, beta):
    """synth.rrc_pulse: unit-energy root-raised cosine, t in symbol periods (float64 tensor)."""
    z = t.abs() < 1e-9
    sg = (t.abs() - 1.0 / (4.0 * beta)).abs() < 1e-9
    ts = torch.where(z | sg, torch.full_like(t, 0.123), t)
    v = (torch.sin(math.pi * ts * (1 - beta)) + 4 * beta * ts * torch.cos(math.pi * ts * (1 + beta))) / (
        math.pi * ts * (1 - (4 * beta * ts) ** 2))
    v = torch.where(z, torch.full_like(t, 1 - beta + 4 * beta / math.pi), v)
    edge = (beta / math.sqrt(2)) * ((1 + 2 / math.pi) * math.sin(math.pi / (4 * beta)) + (1 - 2 / math.pi) * math.cos(math.pi / (4 * beta)))
    return torch.where(sg, torch.full_like(t, edge), v)


def modulate_batch(torch, device, bits, n_samples, tau, cfo, amp, phase0, esn0_db=25.0, noise_seed=0, beta=0.35, chunk=128):
    """bits uint8 [C][2K] (tensor on `device`), per-channel float64 tensors tau / cfo / amp / phase0 [C] -> complex64 [C][n_samples]:
    synth.gen_channel's signal at 2 samples per symbol (sample n at symbol time (n + tau) / 2 - SPAN, the pulse truncated to +-SPAN
    symbols), AWGN at Es/N0 = esn0_db from a generator seeded with noise_seed (None: no noise)."""
    SPAN = synth.SPAN
    C, nb = bits.shape
    K = nb // 2
    assert n_samples % 2 == 0
    M = n_samples // 2
    out = torch.empty((C, n_samples), dtype=torch.complex64, device=device)
    lut = torch.tensor([1, 3, -1, -3], dtype=torch.int64, device=device)          # dibit (first bit << 1 | second) -> step in pi / 4
    g = None
    if esn0_db is not None:
        g = torch.Generator(device=device)
        g.manual_seed(int(noise_seed))
    j = torch.arange(-SPAN, SPAN + 1, dtype=torch.int64, device=device)
    n = torch.arange(n_samples, dtype=torch.float64, device=device)
    m = torch.arange(M, dtype=torch.int64, device=device)
    front = 2 * SPAN + 2
    for c0 in range(0, C, chunk):
        c1 = min(C, c0 + chunk)
        b = bits[c0:c1].to(torch.int64)
        ph = torch.cumsum(lut[(b[:, 0::2] << 1) | b[:, 1::2]], dim=1) % 8
        syms = torch.polar(torch.ones_like(ph, dtype=torch.float64), ph.to(torch.float64) * (math.pi / 4))
        pad = torch.zeros((c1 - c0, K + 2 * front + 2), dtype=torch.complex128, device=device)
        pad[:, front:front + K] = syms
        acc = torch.zeros((c1 - c0, M, 2), dtype=torch.complex128, device=device)
        for p in range(2):                                                        # the two samples of a symbol period
            tp = (p + tau[c0:c1]) / 2.0                                           # [c]: symbol time of sample p, before the -SPAN
            kk = torch.floor(tp)
            fr = tp - kk                                                          # fractional symbol offset of every sample 2 m + p
            taps = _rrc(torch, fr[:, None] - j[None, :].to(torch.float64), beta)  # [c][2 SPAN + 1]
            base = m[None, :] + kk.to(torch.int64)[:, None] - SPAN + front        # symbol index of tap j = 0, padded
            for ji in range(2 * SPAN + 1):
                idx = (base + int(j[ji])).clamp(0, pad.shape[1] - 1)
                acc[:, :, p] += torch.gather(pad, 1, idx) * taps[:, ji][:, None]
        s = acc.reshape(c1 - c0, n_samples)
        rot = torch.polar(amp[c0:c1, None].expand(-1, n_samples).contiguous(), cfo[c0:c1, None] * n[None, :] + phase0[c0:c1, None])
        s = s * rot
        if g is not None:
            sigma = torch.sqrt(amp[c0:c1] * amp[c0:c1] * 2.0 / (10.0 ** (esn0_db / 10.0)) / 2.0)
            s = s + sigma[:, None] * torch.view_as_complex(torch.randn((c1 - c0, n_samples, 2), dtype=torch.float64, device=device, generator=g))
        out[c0:c1] = s.to(torch.complex64)
    return out


def gen_bank(torch, device, n_channels, n_samples, base_seed, esn0_db=25.0, chunk=128):
    """A bank of n_channels independent channels, channel c from seed base_seed + c (bits = synth.hash_bits(seed, ..), parameters =
    synth.hash_params(seed)).  Returns (iq complex64 [C][n_samples] on device, seeds numpy int64 [C])."""
    seeds = np.arange(n_channels, dtype=np.int64) + int(base_seed)
    nb = synth.needed_bits(n_samples)
    out = torch.empty((n_channels, n_samples), dtype=torch.complex64, device=device)
    for c0 in range(0, n_channels, 1024):                    # (bits of 1024 channels at a time: 1024 x 36 k x 8 B as int64)
        c1 = min(n_channels, c0 + 1024)
        sd = torch.as_tensor(seeds[c0:c1], device=device)
        bits = hash_bits(torch, device, sd, nb)
        prm = hash_params(torch, device, sd)
        out[c0:c1] = modulate_batch(torch, device, bits, n_samples, prm["tau"], prm["cfo"], prm["amp"], prm["phase0"], esn0_db=esn0_db,
                                    noise_seed=int(base_seed) + c0, chunk=chunk)
    return out, seeds
