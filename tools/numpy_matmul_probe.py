#!/usr/bin/env python3
"""Which floating-point association does numpy's matmul use for the reference's camera matrices?

The reference computes `E @ car.get_3d_transformation_matrix()`, `(extrinsic @ points.T).T` and `K @ points.T`
(tinycarlo/camera.py:62,131,138, car.py:165) with numpy, i.e. OpenBLAS dgemm, and `R_M.dot([tx, ty])` (car.py:111)
with dgemv.  This script replays such products with exact rational arithmetic under several candidate associations
(unfused ascending, one fused-multiply-add chain ascending from zero, the same descending, two accumulators) and
prints which one reproduces numpy bit for bit.  Result in the build container (numpy 2.2 + OpenBLAS 0.3.29, x86-64 with
FMA): dgemm = ascending FMA chain from a zero accumulator, 100 % of entries; dgemv 2x2 = fma(R[i][0], x0, R[i][1]*x1),
100 %.  oracle/tc_oracle.c (matmul, orc_car_step) and tinycarlo_amd/csrc/tc_device.h (d_matmul, d_car_step) follow it.
CPU only, ~10 s.
"""
import math
from fractions import Fraction as Fr

import numpy as np


def fma(a, b, c):
    return float(Fr(float(a)) * Fr(float(b)) + Fr(float(c)))  # exact product and sum, ONE rounding


def mm(A, B, scheme):
    n, k = A.shape
    p = B.shape[1]
    C = np.zeros((n, p))
    for i in range(n):
        for j in range(p):
            if scheme == "unfused ascending":
                acc = A[i, 0] * B[0, j]
                for t in range(1, k):
                    acc = acc + A[i, t] * B[t, j]
            elif scheme == "fma chain ascending from 0":
                acc = 0.0
                for t in range(k):
                    acc = fma(A[i, t], B[t, j], acc)
            elif scheme == "fma chain descending from 0":
                acc = 0.0
                for t in range(k - 1, -1, -1):
                    acc = fma(A[i, t], B[t, j], acc)
            else:  # two accumulators (even / odd k)
                a0 = a1 = 0.0
                for t in range(0, k, 2):
                    a0 = fma(A[i, t], B[t, j], a0)
                for t in range(1, k, 2):
                    a1 = fma(A[i, t], B[t, j], a1)
                acc = a0 + a1
            C[i, j] = acc
    return C


def main():
    rng = np.random.default_rng(0)
    schemes = ["unfused ascending", "fma chain ascending from 0", "fma chain descending from 0", "two fma accumulators"]
    score = {s: [0, 0] for s in schemes}
    for trial in range(40):
        th, x, y = rng.uniform(-math.pi, math.pi), rng.uniform(0, 2.2), rng.uniform(0, 1.5)
        R = np.array([[math.cos(th), -math.sin(th), 0, 0], [math.sin(th), math.cos(th), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
        T = np.array([[1, 0, 0, -x], [0, 1, 0, -y], [0, 0, 1, 0], [0, 0, 0, 1.0]])
        E = rng.normal(size=(3, 4))
        K = np.array([[38.136, 0, 32], [0, 38.136, 32], [0, 0, 1.0]])
        pts = np.column_stack((rng.uniform(0, 2.2, (24, 2)), np.zeros(24), np.ones(24)))
        car3d = R @ T
        pose = E @ car3d
        P = (pose @ pts.T).T
        H = K @ P.T
        for s in schemes:
            for got, ref in ((mm(R, T, s), car3d), (mm(E, car3d, s), pose), (mm(pose, pts.T, s).T, P), (mm(K, P.T, s), H)):
                score[s][0] += int((got == ref).sum())
                score[s][1] += ref.size
    print("dgemm (A @ B):")
    for s in schemes:
        print(f"  {s:32s} {score[s][0]:6d} / {score[s][1]} entries bit-equal to numpy")
    cnt = {}
    N = 5000
    for _ in range(N):
        dy, tx, ty = rng.uniform(-0.2, 0.2), rng.uniform(-1, 1), rng.uniform(-1, 1)
        Rm = np.array([[math.cos(dy), -math.sin(dy)], [math.sin(dy), math.cos(dy)]])
        r = Rm.dot([tx, ty])
        a, b, c, d = Rm[0, 0], Rm[0, 1], Rm[1, 0], Rm[1, 1]
        cands = {"unfused a*x0 + b*x1": (a * tx + b * ty, c * tx + d * ty),
                 "fma(b, x1, a*x0)": (fma(b, ty, a * tx), fma(d, ty, c * tx)),
                 "fma(a, x0, b*x1)": (fma(a, tx, b * ty), fma(c, tx, d * ty))}
        for k, v in cands.items():
            cnt[k] = cnt.get(k, 0) + int(v[0] == r[0] and v[1] == r[1])
    print("dgemv (2x2 matrix . vector):")
    for k, v in cnt.items():
        print(f"  {k:32s} {v:6d} / {N} results bit-equal to numpy")


if __name__ == "__main__":
    main()
