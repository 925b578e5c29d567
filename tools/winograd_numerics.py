#!/usr/bin/env python3
"""f32 accuracy of the Winograd / Toom-Cook forms considered for the convolutions (CPU, numpy; no GPU needed).

Builds A^T, G, B^T exactly (fractions) for F(m, r) at the usual points, runs the whole pipeline in float32 -- weights
transformed in f64 and rounded once (as the pack kernels do), input transform, 256-channel accumulation, output transform --
and compares with a float64 direct correlation; a serial float32 direct sum is the yardstick.  This is where DESIGN.md's
numbers come from: 1-D F(4,5) (the GRU kernels) costs about as much accuracy as a different summation order, 2-D F(4x4,3x3)
twelve times F(2x2,3x3)'s error (so the 3x3 layers stay at F(2x2)).
    python tools/winograd_numerics.py
"""
import math
from fractions import Fraction as Fr

import numpy as np


def matrices(points, m, r):
    """y = A^T [(G g) . (B^T d)] for m outputs of an r-tap correlation; points = the finite evaluation points (+ infinity)."""
    n = m + r - 1

    def E(cols):
        return [[Fr(a) ** k for k in range(cols)] for a in points] + [[Fr(0)] * (cols - 1) + [Fr(1)]]
    En = E(n)
    A = [row[:] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(En)]
    for c in range(n):                                        # exact inverse of the n x n evaluation matrix
        p = next(i for i in range(c, n) if A[i][c] != 0); A[c], A[p] = A[p], A[c]
        pv = A[c][c]; A[c] = [x / pv for x in A[c]]
        for i in range(n):
            if i != c and A[i][c] != 0:
                f = A[i][c]; A[i] = [x - f * y for x, y in zip(A[i], A[c])]
    C = [row[n:] for row in A]
    BT = [[C[j][i] for j in range(n)] for i in range(n)]
    G = E(r)
    AT = [[E(m)[j][i] for j in range(n)] for i in range(m)]
    for j in range(n):                                        # integer B^T rows, the scale moved into G
        den = 1
        for x in BT[j]: den = den * x.denominator // math.gcd(den, x.denominator)
        num = 0
        for x in BT[j]: num = math.gcd(num, int(x * den))
        s = Fr(den, num)
        BT[j] = [x * s for x in BT[j]]; G[j] = [x / s for x in G[j]]
    return tuple(np.array([[float(x) for x in row] for row in M]) for M in (AT, G, BT))


def run_1d(m, r, points, cin=256, trials=400, seed=0):
    AT, G, BT = matrices(points, m, r)
    n = m + r - 1
    rng = np.random.default_rng(seed)
    g = (rng.standard_normal((cin, r)) * 0.05).astype(np.float32)
    U = (G @ g.T.astype(np.float64)).astype(np.float32)
    ew, ed = [], []
    for _ in range(trials):
        d = (rng.standard_normal((cin, n)) * 2).astype(np.float32)
        ref = np.array([(g.astype(np.float64) * d[:, i:i + r].astype(np.float64)).sum() for i in range(m)])
        V = (BT.astype(np.float32) @ d.T).astype(np.float32)
        M = np.zeros(n, np.float32)
        for c in range(cin): M = (M + U[:, c] * V[:, c]).astype(np.float32)
        y = (AT.astype(np.float32) @ M).astype(np.float32)
        dd = np.zeros(m, np.float32)
        for c in range(cin):
            for k in range(r): dd = (dd + g[c, k] * d[c, np.arange(m) + k]).astype(np.float32)
        ew.append(np.abs(y - ref).max()); ed.append(np.abs(dd - ref).max())
    return max(ew), float(np.mean(ew)), max(ed), float(np.mean(ed)), n / float(r * m)


def run_2d(m, points, cin=256, trials=150, seed=0):
    AT, G, BT = matrices(points, m, 3)
    n = m + 2
    rng = np.random.default_rng(seed)
    g = (rng.standard_normal((cin, 3, 3)) * 0.05).astype(np.float32)
    U = np.einsum('ij,cjk,lk->cil', G, g.astype(np.float64), G).astype(np.float32)
    B32, A32 = BT.astype(np.float32), AT.astype(np.float32)
    ew, ed = [], []
    for _ in range(trials):
        d = (rng.standard_normal((cin, n, n)) * 2).astype(np.float32)
        ref = np.array([[(g.astype(np.float64) * d[:, i:i + 3, j:j + 3]).sum() for j in range(m)] for i in range(m)])
        V = np.einsum('cik,lk->cil', np.einsum('ij,cjk->cik', B32, d).astype(np.float32), B32).astype(np.float32)
        M = np.zeros((n, n), np.float32)
        for c in range(cin): M = (M + U[c] * V[c]).astype(np.float32)
        y = ((A32 @ M).astype(np.float32) @ A32.T).astype(np.float32)
        dd = np.zeros((m, m), np.float32)
        for c in range(cin):
            for a in range(3):
                for b in range(3): dd = (dd + g[c, a, b] * d[c, a:a + m, b:b + m]).astype(np.float32)
        ew.append(np.abs(y - ref).max()); ed.append(np.abs(dd - ref).max())
    return max(ew), float(np.mean(ew)), max(ed), float(np.mean(ed)), n * n / (9.0 * m * m)


if __name__ == '__main__':
    H = Fr(1, 2)
    print('256-channel sums, inputs ~N(0, 2), weights ~N(0, 0.05): |error| against f64   (max, mean | serial direct f32: max, mean | products per output)')
    for m, pts in ((2, (0, 1, -1, 2, -2)), (3, (0, 1, -1, 2, -2, H)), (4, (0, 1, -1, 2, -2, H, -H)), (6, (0, 1, -1, 2, -2, H, -H, 3, -3))):
        print('1-D F(%d,5)      winograd %.2e %.2e | direct %.2e %.2e | %.3f' % ((m,) + run_1d(m, 5, pts)))
    for m, pts in ((2, (0, 1, -1)), (3, (0, 1, -1, 2)), (4, (0, 1, -1, 2, -2))):
        print('2-D F(%dx%d,3x3)  winograd %.2e %.2e | direct %.2e %.2e | %.3f' % ((m, m) + run_2d(m, pts)))
