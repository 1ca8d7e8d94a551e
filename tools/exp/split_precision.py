#!/usr/bin/env python3
"""Numerical price of operand splits for fp32 products on 16-bit matrix instructions (CPU, numpy; float64 reference):
  bf16x3 / 6 terms  — today's convs (x = x1 + x2 + x3 exactly, six products kept);
  fp16x2 / 3 terms  — x = s (h1 + h2), s a power of two per tensor, three products kept: HALF the matrix work and 4 instead of
                       6 bytes per pre-split element, 22 instead of 24 mantissa bits;
  fp32 sequential    — a float32 CPU convolution's accumulation.
Reductions of K = 576 / 4608 terms (3x3 convs at 64 / 512 channels), operands like a training step's: ReLU'd activations x
He-initialised weights (forward), and a gradient tensor with a wide dynamic range x activations (weight gradient)."""
import numpy as np


def bf16_round(a):
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split_bf16x3(a):
    a = a.astype(np.float32)
    p1 = bf16_round(a); r = a - p1
    p2 = bf16_round(r); r = r - p2
    return p1, p2, bf16_round(r)


def split_fp16x2(a):
    a = a.astype(np.float32)
    s = np.float32(2.0) ** np.floor(np.log2(np.abs(a).max() / 2.0 ** 14))       # largest element -> [2^14, 2^15)
    b = a / s
    h1 = b.astype(np.float16).astype(np.float32)
    h2 = (b - h1).astype(np.float16).astype(np.float32)
    return s, h1, h2


def acc32(terms):
    """sum over the last axis in float32, sequentially in chunks of 16 (one matrix-instruction K step), chunk sums exact-ish"""
    t = terms.astype(np.float32)
    k = t.shape[-1]
    out = np.zeros(t.shape[:-1], np.float32)
    for i in range(0, k, 16):
        out = out + t[..., i:i + 16].sum(-1, dtype=np.float32)
    return out


def run(name, x, y):
    ref = (x.astype(np.float64) * y.astype(np.float64)).sum(-1)
    scale = np.abs(ref).max()
    rows = []
    rows.append(("fp32 sequential", acc32(x.astype(np.float32) * y.astype(np.float32))))
    a, b = split_bf16x3(x), split_bf16x3(y)
    six = acc32(a[0] * b[0]) + acc32(a[0] * b[1]) + acc32(a[1] * b[0]) + acc32(a[0] * b[2]) + acc32(a[2] * b[0]) + acc32(a[1] * b[1])
    rows.append(("bf16x3, 6 terms", six))
    (sx, x1, x2), (sy, y1, y2) = split_fp16x2(x), split_fp16x2(y)
    three = (acc32(x1 * y1) + acc32(x1 * y2) + acc32(x2 * y1)) * (sx * sy)
    rows.append(("fp16x2, 3 terms", three))
    print(f"{name}: K = {x.shape[-1]}, {x.shape[0]} dot products, max |ref| = {scale:.3e}")
    for label, got in rows:
        err = np.abs(got.astype(np.float64) - ref)
        print(f"   {label:18s} max err / max|ref| = {err.max() / scale:.2e}   rms err / rms ref = {np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean()):.2e}")


def main():
    rs = np.random.RandomState(0)
    for k in (576, 4608):
        x = np.maximum(rs.randn(4096, k), 0).astype(np.float32)
        w = (rs.randn(4096, k) * np.sqrt(2.0 / k)).astype(np.float32)
        run("forward (activations x weights)", x, w)
        dy = (rs.randn(4096, k) * np.exp(rs.randn(4096, k) * 2.0) * 1e-4).astype(np.float32)      # heavy-tailed gradient
        run("weight gradient (gradients x activations)", dy, x)


if __name__ == "__main__":
    main()
