"""Bit-exact numpy mirror of the stateless RNG in llm_bci_amd/csrc/nbci_common.h.

The reference draws dropout/noise from PyTorch's Philox streams (torch.nn.Dropout,
torch.randn: models/ndt1.py:99-104,158,222,256,289), which no other implementation can
reproduce; parity for the stochastic parts is therefore defined against THIS generator,
which the HIP kernels use, so train-mode runs of the HIP path and the oracle see identical
masks.
"""
import numpy as np

U32 = np.uint32


def mix32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> U32(16); x *= U32(0x7FEB352D)
        x ^= x >> U32(15); x *= U32(0x846CA68B)
        x ^= x >> U32(16)
    return x


def rng_u32(seed, site, idx):
    idx = np.asarray(idx, dtype=np.uint32)
    with np.errstate(over="ignore"):
        s1 = U32((int(seed) * 0x9E3779B9 + 0x85EBCA6B) & 0xFFFFFFFF)
        s2 = U32((int(site) * 0xC2B2AE35 + 0x27D4EB2F) & 0xFFFFFFFF)
    return mix32(mix32(idx ^ s1) ^ s2)


def drop_key(seed, site):
    return int(mix32(U32((int(seed) * 0x9E3779B9 + int(site) * 0x85EBCA6B + 0x27D4EB2F) & 0xFFFFFFFF)))


def drop_threshold(p):
    """16-bit threshold: P(drop) = floor(p * 65536) / 65536."""
    t = float(np.float32(p)) * 65536.0
    if t <= 0:
        return 0
    if t >= 65535.0:
        return 65535
    return int(t)


def keep_mask(seed, site, n, p):
    """float32 array of n multipliers: 1/(1-p) where kept, 0 where dropped (p == 0 -> ones).
    One 32-bit hash per PAIR of elements, low 16 bits for the even index, high 16 for the odd."""
    thr = drop_threshold(p)
    if thr == 0:
        return np.ones(n, np.float32)
    idx = np.arange(n, dtype=np.uint32)
    h = mix32((idx >> U32(1)) ^ U32(drop_key(seed, site)))
    draws = np.where((idx & U32(1)) == 1, h >> U32(16), h & U32(0xFFFF))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(draws >= U32(thr), scale, np.float32(0)).astype(np.float32)


def normal(seed, site, n):
    """Box-Muller normals matching rng_normal() up to the GPU's fast log/cos (|diff| ~1e-6)."""
    idx = np.arange(n, dtype=np.uint32)
    a = rng_u32(seed, site, idx)
    b = rng_u32((int(seed) ^ 0x5BD1E995) & 0xFFFFFFFF, (int(site) + 0x1000193) & 0xFFFFFFFF, idx)
    u1 = ((a >> U32(8)).astype(np.float32) + np.float32(1)) * np.float32(1.0 / 16777216.0)
    u2 = ((b >> U32(8)).astype(np.float32) + np.float32(1)) * np.float32(1.0 / 16777216.0)
    return (np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)).astype(np.float32)


SITE_NOISE_WHITE = 1


def white_noise(seed, B, T, N):
    """(B,T,N) standard normals as the smoothing kernel draws them: bins (t, t+1) of one (batch, channel)
    share one Box-Muller transform (cos -> even bin, sin -> odd bin); pair index =
    (b*N + n) * ceil(T/2) + t//2, site 1."""
    npair = (T + 1) // 2
    idx = np.arange(B * N * npair, dtype=np.uint32)
    a = rng_u32(seed, SITE_NOISE_WHITE, idx)
    b = rng_u32((int(seed) ^ 0x5BD1E995) & 0xFFFFFFFF, (SITE_NOISE_WHITE + 0x1000193) & 0xFFFFFFFF, idx)
    u1 = ((a >> U32(8)).astype(np.float32) + np.float32(1)) * np.float32(1.0 / 16777216.0)
    u2 = ((b >> U32(8)).astype(np.float32) + np.float32(1)) * np.float32(1.0 / 16777216.0)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1))
    ang = np.float32(6.283185307179586) * u2
    z = np.stack([rad * np.cos(ang), rad * np.sin(ang)], -1).reshape(B, N, 2 * npair)[:, :, :T]
    return np.ascontiguousarray(z.transpose(0, 2, 1)).astype(np.float32)


# dropout / noise call sites of one NDT1 train step (shared numbering with csrc/ndt1.hip)
SITE_NOISE_OFFSET = 2
SITE_EMBED_DROP = 3


def site_attn_prob(layer):
    return 16 + 4 * layer


def site_attn_out(layer):
    return 17 + 4 * layer


def site_mlp_out(layer):
    return 18 + 4 * layer
