"""numpy model of the MX-scaled fp8 (OCP e4m3) quantisation the HIP path applies to PatchTST's q / k / v projections (csrc/fp8.hip,
BASELINE configs[4] "fp8 MFMA QKV"). Test infrastructure only - see oracle/__init__.py.

The arithmetic is not the reference's (it runs these Linears in fp32 / autocast bf16, models/patchtst.py:176 -> HF PatchTSTModel): it
is the OCP Microscaling (MX) v1.0 format restated - MXFP8 with E4M3 elements: blocks of 32 consecutive values share a power-of-two
scale X = 2^(floor(log2 amax) - 8) (8 = the largest e4m3 exponent), one step larger when amax / X would exceed 448 (the plain rule
saturates block maxima in (1.75, 2) x 2^n), stored as the biased byte E + 127 (amax = 0 or subnormal -> 0);
elements = round-to-nearest-even e4m3 of v / X (a clamp to +-448 remains as a guard). e4m3 (OCP "fn"): 1 sign, 4 exponent (bias 7), 3 mantissa bits, no
infinities, subnormals at exponent field 0 (step 2^-9), largest finite 448. What the tests pin: the device's codes and scale bytes
equal these bit for bit, and its GEMM equals the f32 product of the dequantised operands.
"""
import numpy as np


def e4m3_round(v):
    """round-to-nearest-even onto the e4m3 grid, saturating at +-448; f32 in, f32 (exact grid values) out."""
    v = np.asarray(v, np.float32)
    a = np.minimum(np.abs(v), np.float32(448.0)).astype(np.float64)
    e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.maximum(e, -6.0)                                   # below 2^-6: the subnormal step 2^-9
    step = np.exp2(e - 3.0)
    q = np.rint(a / step) * step                              # np.rint = round half to even
    q = np.minimum(q, 448.0)
    return np.copysign(q, v).astype(np.float32)               # (keeps the sign of a zero, as the hardware conversion does)


def e4m3_encode(v):
    """the e4m3 byte of a value already on the grid (for bit-level comparison with the device's codes)."""
    v = np.asarray(v, np.float32)
    a = np.abs(v).astype(np.float64)
    s = (np.signbit(v)).astype(np.uint8) << 7
    e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    normal = a >= 2.0 ** -6
    be = np.where(normal, e + 7, 0).astype(np.int64)
    man = np.where(normal, np.rint((a / np.exp2(e) - 1.0) * 8.0), np.rint(a / 2.0 ** -9)).astype(np.int64)
    return (s | (be.astype(np.uint8) << 3) | man.astype(np.uint8)).astype(np.uint8)


def mx_quantize(x):
    """x (..., K), K % 32 == 0 -> (dequantised f32 values, e4m3 codes uint8 (..., K), E8M0 scale bytes uint8 (..., K/32))."""
    x = np.asarray(x, np.float32)
    K = x.shape[-1]
    b = x.reshape(x.shape[:-1] + (K // 32, 32))
    amax = np.abs(b).max(-1)
    bits = amax.view(np.uint32)
    ex = ((bits >> 23) & 0xFF).astype(np.int64)
    E = np.maximum(ex - 127 - 8, -127)
    over = (amax.astype(np.float64) * np.exp2(-E.astype(np.float64)) > 448.0) & (E < 127)
    E = E + over                                             # one step up where amax / X would exceed 448: nothing saturates
    sb = np.where(ex == 0, 0, E + 127).astype(np.uint8)
    scale = np.exp2(sb.astype(np.float64) - 127.0)
    q = e4m3_round((b.astype(np.float64) / scale[..., None]).astype(np.float32))
    deq = (q.astype(np.float64) * scale[..., None]).astype(np.float32).reshape(x.shape)
    return deq, e4m3_encode(q).reshape(x.shape), sb


def linear_fp8(x, w, bias=None):
    """y = dequant(mx(x)) . dequant(mx(w))^T + bias in f32 (the device accumulates exact fp8 products in f32)."""
    xd, _, _ = mx_quantize(x)
    wd, _, _ = mx_quantize(w)
    y = xd.reshape(-1, xd.shape[-1]).astype(np.float64) @ wd.astype(np.float64).T
    if bias is not None:
        y = y + np.asarray(bias, np.float64)
    return y.astype(np.float32).reshape(x.shape[:-1] + (w.shape[0],))
