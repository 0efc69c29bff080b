"""numpy emulation of csrc/mlp_reg.hip's register bookkeeping (MFMA 32x32x2 lane layouts, permlane32_swap),
to check the index logic of the transient/persistent layer scheme against a plain matmul chain."""
import numpy as np
rng = np.random.default_rng(0)
L = np.arange(64); J = L & 31; H = L >> 5

def mfma(a, b, acc):            # a, b: [64] registers; acc: [16, 64]
    A = np.zeros((32, 2)); Bm = np.zeros((2, 32))
    A[J, H] = a; Bm[H, J] = b
    D = A @ Bm                  # [oc row, sample col]
    out = acc.copy()
    for reg in range(16):
        row = (reg & 3) + 8 * (reg >> 2) + 4 * H
        out[reg] += D[row, J]
    return out

def swap32(a, b):
    lo = a.copy(); hi = b.copy()
    lo[32:] = b[:32]; hi[:32] = a[32:]
    return lo, hi

def to_operands(v0, v1, v2, v3):
    s01 = swap32(v0, v1); s23 = swap32(v2, v3)
    return [s01[0], s23[0], s01[1], s23[1]]

def pack(W, KP, NP):            # fragments [oc tile][t][lane][e]
    Cout, Cin = W.shape
    nT = KP // 8
    F = np.zeros((NP // 32, nT, 64, 4))
    for oct_ in range(NP // 32):
        for t in range(nT):
            for e in range(4):
                oc = oct_ * 32 + J; k = 8 * t + 2 * e + H
                ok = (oc < Cout) & (k < Cin)
                F[oct_, t, :, e] = np.where(ok, W[np.minimum(oc, Cout - 1), np.minimum(k, Cin - 1)], 0.0)
    return F

def bias_tile(b, o):
    t = np.zeros((16, 64))
    for a in range(4):
        for q in range(4):
            t[4 * a + q] = b[o * 32 + 8 * a + 4 * H + q]
    return t

dims = [8, 64, 64, 128]         # padded input of 8 channels
X = rng.normal(size=(32, 8))    # 32 rows
Ws = [rng.normal(size=(dims[i + 1], dims[i])) for i in range(3)]
bs = [rng.normal(size=(dims[i + 1],)) for i in range(3)]
ref = X
for W, b in zip(Ws, bs):
    ref = np.maximum(ref @ W.T + b, 0)
NT0, NO0, NG1, NO1, NG2, NO2 = 1, 2, 8, 2, 8, 4
F = [pack(Ws[0], 8, 64), pack(Ws[1], 64, 64), pack(Ws[2], 64, 128)]
# layer-0 operands: lane (j,h) holds channels 4h..4h+3 of row j
in0 = []
for t in range(NT0):
    v = [X[J, 8 * t + 4 * H + i] for i in range(4)]
    in0 += to_operands(*v)
acc1 = [bias_tile(bs[1], o1) for o1 in range(NO1)]
for o in range(NO0):
    t_ = bias_tile(bs[0], o)
    for t in range(NT0):
        for e in range(4):
            t_ = mfma(F[0][o, t, :, e], in0[4 * t + e], t_)
    t_ = np.maximum(t_, 0)
    bt = []
    for a in range(4):
        bt += to_operands(t_[4 * a], t_[4 * a + 1], t_[4 * a + 2], t_[4 * a + 3])
    for i in range(NO1 * 4):
        o1, a = i >> 2, i & 3
        for e in range(4):
            acc1[o1] = mfma(F[1][o1, 4 * o + a, :, e], bt[4 * a + e], acc1[o1])
in2 = [None] * (NG2 * 4)
for o1 in range(NO1):
    t_ = np.maximum(acc1[o1], 0)
    for a in range(4):
        ops = to_operands(t_[4 * a], t_[4 * a + 1], t_[4 * a + 2], t_[4 * a + 3])
        for e in range(4):
            in2[16 * o1 + 4 * a + e] = ops[e]
out = np.zeros((32, 128))
for o in range(NO2):
    t_ = bias_tile(bs[2], o)
    for t in range(NG2):
        for e in range(4):
            t_ = mfma(F[2][o, t, :, e], in2[4 * t + e], t_)
    t_ = np.maximum(t_, 0)
    for reg in range(16):
        ch = o * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * H
        out[J, ch] = t_[reg]
print("max |emulated - reference| =", np.abs(out - ref).max())
