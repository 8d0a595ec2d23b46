#!/usr/bin/env python
"""CPU-only numerics probe for VERDICT r05 next #1 (Winograd F(2x2, 3x3) on the split-word bf16x3 arithmetic): how far from the exact
result do (a) the direct 3-term-split convolution the halo kernels compute and (b) its Winograd form sit, on one SPADE-decoder-sized
layer (C input channels, 3x3, stride 1, pad 1)?  Emulation: every product is a_hi*b_hi + a_hi*b_lo + a_lo*b_hi of bf16 values (exact
in fp32), accumulated in fp32 -- what v_mfma_f32_16x16x32_bf16 does up to summation order.  Winograd: U = G g G^T from the fp32
weights in fp32, split; V = B^T d B in fp32 from the split words' exact values (hi + lo), split again; M_p = U_p V_p; Y = A^T M A in fp32.

    python tools/winograd_numerics.py [C] [K] [side]
"""
import sys

import numpy as np


def bf16_round(x):
    """Round fp32 -> bf16 (RNE), returned as fp32."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split(x):
    hi = bf16_round(x)
    lo = bf16_round((x - hi).astype(np.float32))
    return hi, lo


def mm3(a, b):
    """[M, K] x [K, N] with the 3-term split, fp32 accumulate."""
    ah, al = split(a)
    bh, bl = split(b)
    return (ah @ bh + ah @ bl + al @ bh).astype(np.float32)


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    rs = np.random.RandomState(0)
    x = rs.randn(C, S, S).astype(np.float32)
    x = np.maximum(x, 0.2 * x)                                  # LeakyReLU'd batch-normalised activations
    xh, xl = split(x)
    x = (xh + xl).astype(np.float32)                            # the split words' exact value
    w = (rs.randn(K, C, 3, 3) * 0.02).astype(np.float32)
    xp = np.zeros((C, S + 2, S + 2), np.float32)
    xp[:, 1:-1, 1:-1] = x
    # exact result in fp64
    cols = np.stack([xp[:, r:r + S, s:s + S] for r in range(3) for s in range(3)], 1).reshape(C * 9, S * S)
    ref = w.reshape(K, C * 9).astype(np.float64) @ cols.astype(np.float64)
    # (a) direct, 3-term split
    direct = mm3(w.reshape(K, C * 9), cols)
    # (b) Winograd F(2x2, 3x3)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
    U = np.einsum('ir,kcrs,js->ijkc', G, w, G).astype(np.float32)                                 # [4, 4, K, C]
    T = S // 2
    d = np.stack([np.stack([xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4] for tx in range(T)], 1) for ty in range(T)], 1)   # [C, T, T, 4, 4]
    V = np.einsum('ia,ctuab,jb->ijctu', Bt, d, Bt).astype(np.float32).reshape(4, 4, C, T * T)
    M = np.stack([np.stack([mm3(U[i, j], V[i, j]) for j in range(4)]) for i in range(4)])         # [4, 4, K, T*T]
    Y = np.einsum('ai,ijkt,bj->ktab', At, M, At).astype(np.float32).reshape(K, T, T, 2, 2)
    wino = Y.transpose(0, 1, 3, 2, 4).reshape(K, S * S)
    # (c) the same Winograd form in plain fp32 (no split): what the transform itself costs
    M32 = np.stack([np.stack([(U[i, j] @ V[i, j]).astype(np.float32) for j in range(4)]) for i in range(4)])
    w32 = np.einsum('ai,ijkt,bj->ktab', At, M32, At).astype(np.float32).reshape(K, T, T, 2, 2).transpose(0, 1, 3, 2, 4).reshape(K, S * S)
    d32 = (w.reshape(K, C * 9) @ cols).astype(np.float32)
    print('C %d K %d %dx%d; errors against fp64 as (max / max|y|, relative L2)' % (C, K, S, S))
    for name, y in (('direct fp32', d32), ('direct bf16x3 (the halo kernels)', direct), ('Winograd F(2x2,3x3) fp32', w32), ('Winograd F(2x2,3x3) bf16x3', wino)):
        e = y.astype(np.float64) - ref
        print('  %-34s %.2e  %.2e' % (name, np.abs(e).max() / np.abs(ref).max(), np.linalg.norm(e) / np.linalg.norm(ref)))


if __name__ == '__main__':
    main()
