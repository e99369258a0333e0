import numpy as np, sys
LD = np.longdouble
def ge_solve(A, Bm):  # longdouble Gaussian elimination with partial pivoting: solve A X = Bm
    A = A.astype(LD).copy(); X = Bm.astype(LD).copy(); n = A.shape[0]
    for k in range(n):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]] = A[[p, k]]; X[[k, p]] = X[[p, k]]
        for i in range(k + 1, n):
            f = A[i, k] / A[k, k]
            A[i, k:] -= f * A[k, k:]; X[i] -= f * X[k]
    for k in range(n - 1, -1, -1):
        X[k] = (X[k] - A[k, k + 1:] @ X[k + 1:]) / A[k, k]
    return X
def load(i):
    return np.load(f"/tmp/state_gaussian_blur_{i}.npz")
def apply(D, r, B, M, z):
    t = B @ (r * z)
    return D * z + r * (B.T @ (M @ t))
def step(a, b, T):
    A, Bn = load(a), load(b)
    s = float(np.float32(Bn["sigma"] ** -2.0 - A["sigma"] ** -2.0))
    z = A["z"]
    D, r, B, M = (A[k].astype(T) for k in ("D", "r", "B", "M"))
    m = int(A["m"])
    e = 1 / (1 + T(s) * D)
    G = (B * (r * r * e)) @ B.T
    K = np.eye(m, dtype=T) + T(s) * (G @ M)
    # M' = M K^-1  <=> K^T M'^T = M^T
    if T is LD:
        Mp = ge_solve(K.T, M.T).T
    else:
        Mp = np.linalg.solve(K.T, M.T).T
    out = apply(D * e, r * e, B, Mp, z.astype(T))
    return out, s, Mp, G, K
for a, b in ((18, 20), (20, 21)):
    A, Bn = load(a), load(b)
    truth, s, MpL, GL, KL = step(a, b, LD)
    f64, _, Mp64, G64, K64 = step(a, b, np.float64)
    sc = float(np.abs(truth).max())
    print(f"step {a}->{b}: sigma {float(A['sigma']):.4g} -> {float(Bn['sigma']):.4g}, s = {s:.6g}, m = {int(A['m'])}")
    print("   cond(K) =", np.linalg.cond(K64), " cond(G) =", np.linalg.cond(G64), " |M| range", np.abs(A['M']).min(), np.abs(A['M']).max())
    print("   numpy f64 emulation vs longdouble truth :", float(np.abs(f64 - truth).max()) / sc)
    print("   hip C' z (state after)  vs truth        :", float(np.abs(Bn["Cz_hip"] - truth).max()) / sc)
    print("   oracle C' z             vs truth        :", float(np.abs(Bn["Cz_orc"] - truth).max()) / sc)
    print("   hip vs oracle                           :", float(np.abs(Bn["Cz_hip"] - Bn["Cz_orc"]).max()) / sc)
    print("   M' hip vs longdouble rel:", float(np.abs(Bn["M"] - MpL).max() / np.abs(MpL).max()))

print("---- variants (float64 emulations vs longdouble truth)")
for a, b in ((18, 20), (20, 21)):
    A, Bn = load(a), load(b)
    truth, s, MpL, GL, KL = step(a, b, LD)
    sc = float(np.abs(truth).max())
    D, r, B, M = (A[k] for k in ("D", "r", "B", "M"))
    z = A["z"]; m = int(A["m"])
    e = 1 / (1 + s * D)
    G = (B * (r * r * e)) @ B.T
    def rel(Mp):
        return float(np.abs(apply(D * e, r * e, B, Mp, z) - truth).max()) / sc
    # 0 plain
    K = np.eye(m) + s * (G @ M)
    print(f"step {a}->{b}  plain LU            : {rel(np.linalg.solve(K.T, M.T).T):.3g}   cond {np.linalg.cond(K):.3g}")
    # 1 balanced by G's diagonal
    S = 1 / np.sqrt(np.diag(G))
    Gt, Mt = G * S[:, None] * S[None, :], M / S[:, None] / S[None, :]
    Kt = np.eye(m) + s * (Gt @ Mt)
    Mp = np.linalg.solve(Kt.T, Mt.T).T * S[:, None] * S[None, :]
    print(f"           balanced LU         : {rel(Mp):.3g}   cond {np.linalg.cond(Kt):.3g}  cond(Gt) {np.linalg.cond(Gt):.3g}")
    # 2 G from longdouble, rest f64 (is the Gram the problem?)
    GLd = GL.astype(np.float64)
    K2 = np.eye(m) + s * (GLd @ M)
    print(f"           exact G, plain LU   : {rel(np.linalg.solve(K2.T, M.T).T):.3g}")
    # 3 symmetric: M' = (M^-1 + s G)^-1, balanced
    try:
        Mi_ = np.linalg.inv(Mt)
        Mp3 = np.linalg.inv(Mi_ + s * Gt) * S[:, None] * S[None, :]
        print(f"           symmetric balanced  : {rel(Mp3):.3g}   cond(Mt) {np.linalg.cond(Mt):.3g}")
    except Exception as ex:
        print("           symmetric failed", ex)
    # 4 eigen-decomposition of Gt: Gt = Q L Q^T, drop nothing;  K = I + s Q L Q^T Mt
    L, Q = np.linalg.eigh(Gt)
    print("           eig(Gt):", " ".join(f"{v:.2e}" for v in L))
    # 5 one step of iterative refinement on M' K = M in f64 with longdouble residual
    Mp0 = np.linalg.solve(K.T, M.T).T
    Rr = (M.astype(LD) - Mp0.astype(LD) @ (np.eye(m, dtype=LD) + LD(s) * (G.astype(LD) @ M.astype(LD)))).astype(np.float64)
    Mp5 = Mp0 + np.linalg.solve(K.T, Rr.T).T
    print(f"           LU + 1 refinement (f64 G): {rel(Mp5):.3g}")
