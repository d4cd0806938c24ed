"""CPU experiment (round 5): can a DEFLATED second Gram pass replace the refinement pass of fit()?

Stage 1: G = X0^T X0 (f64), eigh -> V1, S1 (small modes carry eps (s1/si)^2).  Deflated stage 2: d = modes above thr * s1;
X1 = X0 (I - Vd Vd^T) formed row by row (what a Gram kernel's staging would do), G1 = X1^T X1, eigh -> the modes d+1..r; V = [Vd | V2].
Compared with LAPACK's SVD of X0 (the reference, :272): singular values, right vectors, the basis Ur = X0 V S^-1 (up to sign), and the
existing refinement (Y = X0 V1/S1, H = Y^T Y, X0 = Q M, SVD of M)."""
import numpy as np

rng = np.random.default_rng(0)
n, m, r = 60000, 64, 32


def make(kappa):
    U = np.linalg.qr(rng.standard_normal((n, m)))[0]
    V = np.linalg.qr(rng.standard_normal((m, m)))[0]
    s = kappa ** (-np.arange(m) / (r - 1.0))
    return (U * s) @ V.T


def align(A, B):
    return A * np.sign(np.sum(A * B, axis=0))


def report(tag, S, V, X0, Sr, Vr, Ur):
    U = X0 @ V[:, :r] / S[:r]
    e_s = np.max(np.abs(S[:r] - Sr[:r]) / Sr[:r])
    e_v = np.max(np.linalg.norm(align(V[:, :r], Vr[:, :r]) - Vr[:, :r], axis=0))
    e_u = np.max(np.linalg.norm(align(U, Ur[:, :r]) - Ur[:, :r], axis=0))
    orth = np.abs(U.T @ U - np.eye(r)).max()
    print(f'  {tag:34s} sigma {e_s:8.1e}  V {e_v:8.1e}  Ur {e_u:8.1e}  |Ur^T Ur - I| {orth:8.1e}')


for kappa in (1e4, 1e5, 1e6, 1e7):
    X0 = make(kappa)
    Ur, Sr, Vtr = np.linalg.svd(X0, full_matrices=False)
    Vr = Vtr.T
    print(f'sigma_1/sigma_r = {kappa:g}')
    lam, V1 = np.linalg.eigh(X0.T @ X0)
    lam, V1 = lam[::-1], V1[:, ::-1]
    S1 = np.sqrt(np.maximum(lam, 0))
    report('plain Gram route', S1, V1, X0, Sr, Vr, Ur)
    # existing refinement
    dd = np.maximum(S1, S1[0] * np.sqrt(m * 2.2e-16))
    Y = X0 @ (V1 / dd)
    lh, Z = np.linalg.eigh(Y.T @ Y)
    M = (np.sqrt(np.maximum(lh, 0))[:, None] * Z.T) * dd[None, :] @ V1.T
    _, S2, Vt2 = np.linalg.svd(M)
    report('refinement pass (3 n m^2 flops)', S2, Vt2.T, X0, Sr, Vr, Ur)
    for thr in (1e-2, 1e-3):
        d = int(np.sum(S1 > thr * S1[0]))
        Vd = V1[:, :d]
        X1 = X0 - (X0 @ Vd) @ Vd.T
        l2, V2 = np.linalg.eigh(X1.T @ X1)
        l2, V2 = l2[::-1], V2[:, ::-1]
        V = np.hstack([Vd, V2[:, :m - d]])
        S = np.concatenate([S1[:d], np.sqrt(np.maximum(l2[:m - d], 0))])
        report(f'deflated, thr {thr:g} (d = {d}), 1 stage', S, V, X0, Sr, Vr, Ur)
        # two deflation stages
        d2 = d + int(np.sum(np.sqrt(np.maximum(l2, 0)) > thr * np.sqrt(l2[0])))
        Vd2 = V[:, :d2]
        X2 = X0 - (X0 @ Vd2) @ Vd2.T
        l3, V3 = np.linalg.eigh(X2.T @ X2)
        l3, V3 = l3[::-1], V3[:, ::-1]
        Vb = np.hstack([Vd2, V3[:, :m - d2]])
        Sb = np.concatenate([S[:d2], np.sqrt(np.maximum(l3[:m - d2], 0))])
        report(f'deflated, thr {thr:g} (d = {d}, {d2}), 2 stages', Sb, Vb, X0, Sr, Vr, Ur)
