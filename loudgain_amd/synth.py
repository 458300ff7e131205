"""Synthetic programme material of the benchmark.

Gaussian noise at -20 dBFS RMS whose level steps through
{0, -6, -12, -30, -50, -80} dB every `step_s` seconds (exercises the absolute
gate, the relative gate and the loudness range), plus a 1 s fs/4 sine at 45
degrees phase and 0.9 FS at t = 5 s (inter-sample peak above the sample peak);
every value snapped to the S16 grid k/32768 so that the reference's S16 feed
(scan.c:414) and the f32 path see identical numbers.
"""
import numpy as np

STEPS_DB = (0.0, -6.0, -12.0, -30.0, -50.0, -80.0)


def track_numpy(frames, channels, rate, seed=0, step_s=10.0):
    rng = np.random.Generator(np.random.Philox(key=0x10AD6A1 ^ int(seed)))
    x = rng.standard_normal((frames, channels), dtype=np.float32) * np.float32(0.1)
    gains = (10.0 ** (np.array(STEPS_DB) / 20.0)).astype(np.float32)
    seg = (np.arange(frames) // int(step_s * rate)) % len(STEPS_DB)
    x *= gains[seg][:, None]
    t0, t1 = int(5 * rate), min(frames, int(6 * rate))
    if t1 > t0:
        n = np.arange(t1 - t0)
        x[t0:t1] += (0.9 * np.sin(2 * np.pi * n / 4.0 + np.pi / 4)).astype(np.float32)[:, None]
    return snap_s16_numpy(x)


def snap_s16_numpy(x):
    return (np.clip(np.round(x.astype(np.float64) * 32768.0), -32768, 32767) / 32768.0).astype(np.float32)


def track_torch(frames, channels, rate, seed=0, step_s=10.0, device="cuda", sine=True):
    """Same construction generated directly in HBM (torch's Philox stream; not
    bit-identical to track_numpy -- copy it back to the host to feed the oracle)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(0x10AD6A1 ^ int(seed))
    x = torch.empty((frames, channels), dtype=torch.float32, device=device)
    piece = 1 << 24
    gains = torch.tensor([10.0 ** (d / 20.0) for d in STEPS_DB], dtype=torch.float32, device=device)
    for off in range(0, frames, piece):
        n = min(piece, frames - off)
        blk = torch.randn((n, channels), generator=g, dtype=torch.float32, device=device) * 0.1
        idx = (torch.arange(off, off + n, device=device) // int(step_s * rate)) % len(STEPS_DB)
        blk *= gains[idx][:, None]
        x[off:off + n] = blk
    t0, t1 = int(5 * rate), min(frames, int(6 * rate))
    if t1 > t0 and sine:
        n = torch.arange(t1 - t0, device=device, dtype=torch.float32)
        x[t0:t1] += (0.9 * torch.sin(2 * np.pi * n / 4.0 + np.pi / 4))[:, None]
    x.mul_(32768.0).round_().clamp_(-32768, 32767).div_(32768.0)
    return x


def adversarial_torch(frames, channels, amplitude=0.8, device="cuda"):
    """Worst case for the true-peak pruning: an fs/4 sine sampled on its peaks (0, A, 0, -A, ...
    per channel, the second channel a quarter period later), so that the sample peak equals the
    true peak in every window and no interpolator window can be skipped."""
    import torch
    n = torch.arange(frames, device=device, dtype=torch.int64)
    x = torch.empty((frames, channels), dtype=torch.float32, device=device)
    tab = torch.tensor([0.0, amplitude, 0.0, -amplitude], dtype=torch.float32, device=device)
    for c in range(channels):
        x[:, c] = tab[(n + c) % 4]
    x.mul_(32768.0).round_().clamp_(-32768, 32767).div_(32768.0)
    return x


def limited_torch(frames, channels, rate, seed=0, device="cuda", ceiling=0.8):
    """Loud, heavily limited programme: Gaussian noise (sigma 0.3) plus a bass and a mid tone through a hard
    limiter at `ceiling` -- crest factor ~8 dB, thousands of samples per second sit ON the ceiling, so the
    sample peak is reached everywhere and the true-peak pruning bound (L1 * max|x| of a chunk against the
    track's sample peak) dismisses next to nothing.  What modern pop masters look like to the scanner."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(0x11A17ED ^ int(seed))
    x = torch.empty((frames, channels), dtype=torch.float32, device=device)
    piece = 1 << 24
    for off in range(0, frames, piece):
        n = min(piece, frames - off)
        t = torch.arange(off, off + n, device=device, dtype=torch.float64)
        blk = torch.randn((n, channels), generator=g, dtype=torch.float32, device=device) * 0.3
        for c in range(channels):
            blk[:, c] += (0.15 * torch.sin(2 * np.pi * 110.0 / rate * t + 0.7 * c)
                          + 0.10 * torch.sin(2 * np.pi * 1870.0 / rate * t + 1.3 * c)).to(torch.float32)
        x[off:off + n] = blk.clamp_(-ceiling, ceiling)
    x.mul_(32768.0).round_().clamp_(-32768, 32767).div_(32768.0)
    return x


def limited_numpy(frames, channels, rate, seed=0, ceiling=0.8):
    """numpy twin of limited_torch (same construction, its own random stream) for the oracle-side tests"""
    rng = np.random.Generator(np.random.Philox(key=0x11A17ED ^ int(seed)))
    x = rng.standard_normal((frames, channels), dtype=np.float32) * np.float32(0.3)
    t = np.arange(frames, dtype=np.float64)
    for c in range(channels):
        x[:, c] += (0.15 * np.sin(2 * np.pi * 110.0 / rate * t + 0.7 * c)
                    + 0.10 * np.sin(2 * np.pi * 1870.0 / rate * t + 1.3 * c)).astype(np.float32)
    return snap_s16_numpy(np.clip(x, -ceiling, ceiling))
