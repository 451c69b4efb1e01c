"""CPU model of the PIPELINED lean step (k_sweep_lean pipelined variant): the column sums C^t_j and the
total S_t of the product column x_t = e_t . P'_t in closed form from quantities of column t-1, so that the LDS exchange
of a column leaves the dependent chain of the next one.  Checks the identities the kernel relies on against the plain
recursion (numpy fp64; deviations are rounding only).  Tooling / documentation, not product code.

    x_t(i,j)  = e_t(i,j) * sc_t * (c0 x_{t-1}(i,j) + c1 C^{t-1}_i + c1 C^{t-1}_j + c2 S_{t-1})
    Y^{t-1}_j = sum_i e_t(i,j) x_{t-1}(i,j)                      (accumulated during step t-1, exchanged through LDS)
    Q_b       = sum_{i : bit_t(i) = b} C^{t-1}_i                 (two masked wave totals of step t-1; S_{t-1} = Q_0 + Q_1)
    N_t(j)    = n_0 T_t[0][b_j] + n_1 T_t[1][b_j]                 (per record, two values)
    C^t_j     = sc_t * (c0 Y^{t-1}_j + (c1 C^{t-1}_j + c2 S_{t-1}) N_t(j) + c1 (T_t[0][b_j] Q_0 + T_t[1][b_j] Q_1))
"""
import numpy as np


def run(H=64, C=200, seed=1, zero_cols=()):
    rng = np.random.default_rng(seed)
    bits = rng.random((C, H)) < rng.random((C, 1))
    T = rng.random((C, 2, 2)) * 10.0 ** rng.integers(-8, 1, size=(C, 1, 1))
    T[:, 1, 0] = T[:, 0, 1]
    for z in zero_cols:
        T[z] = 0.0
    d = rng.random(C) * 1e-3 + 1e-6
    r = np.exp(-d / H)
    q = -np.expm1(-d / H) / H
    c0, c1, c2 = r * r, q * r, q * q
    e = lambda t: T[t][bits[t].astype(int)[:, None], bits[t].astype(int)[None, :]]

    # plain recursion (what k_sweep_lean does): column sums of x_{t-1} by summation
    x = e(0) * 1.0
    plain = []
    for t in range(1, C):
        Cs = x.sum(axis=0)
        S = Cs.sum()
        if not S > 0:
            x = np.zeros_like(x); Cs = np.zeros(H); S1 = 1.0
            uj = c0[t] / H ** 2 + c2[t] + 2 * c1[t] * (1.0 / H)
            P = np.full((H, H), uj)
        else:
            S1 = S
            P = c0[t] * x + c1[t] * (Cs[:, None] + Cs[None, :]) + c2[t] * S
        sc = 2.0 ** -(np.frexp(S1)[1])
        P = P * sc
        x = e(t) * P
        plain.append((S, P.copy()))

    # pipelined: C, Q0, Q1 carried; Y exchanged one step ahead
    x = e(0) * 1.0
    Cs = x.sum(axis=0)                                # prime(): by summation, once
    b1 = bits[1]
    Q = [Cs[~b1].sum(), Cs[b1].sum()]
    Y = (e(1) * x).sum(axis=0)                        # partials of step 0 with the NEXT column's emissions
    worst = 0.0
    for t in range(1, C):
        S = Q[0] + Q[1]
        S_plain, P_plain = plain[t - 1]
        assert (S > 0) == (S_plain > 0), t
        if S_plain > 0:
            worst = max(worst, abs(S - S_plain) / S_plain)
        if not S > 0:
            c0e, ujv, S1 = 0.0, np.full(H, c0[t] / H ** 2 + c2[t] + 2 * c1[t] * (1.0 / H)), 1.0
        else:
            c0e, ujv, S1 = c0[t], c1[t] * Cs + c2[t] * S, S
        sc = 2.0 ** -(np.frexp(S1)[1])
        c0s, ujs, c1s = c0e * sc, ujv * sc, c1[t] * sc
        P = c0s * x + ujs[None, :] + (c1s * Cs)[:, None]            # state block (needs only last step's C, S)
        denom = np.abs(P_plain).max()
        worst = max(worst, np.abs(P - P_plain).max() / denom)
        xn = e(t) * P
        # closed form of this column's sums: everything on the right is known BEFORE the state block ran
        bt = bits[t].astype(int)
        eA, eB = T[t][0][bt], T[t][1][bt]
        n1 = bits[t].sum(); n0 = H - n1
        Cn = c0s * Y + ujs * (n0 * eA + n1 * eB) + c1s * (eA * Q[0] + eB * Q[1])
        worst = max(worst, np.abs(Cn - xn.sum(axis=0)).max() / max(xn.sum(), 1e-300))
        if t + 1 < C:
            bn = bits[t + 1]
            Q = [Cn[~bn].sum(), Cn[bn].sum()]
            Y = (e(t + 1) * xn).sum(axis=0)
        Cs, x = Cn, xn
    return worst


if __name__ == "__main__":
    for seed in range(5):
        w = run(seed=seed, zero_cols=(37, 38, 90) if seed % 2 else ())
        print("seed", seed, "worst relative deviation pipelined vs plain:", w)
        assert w < 1e-12
    print("OK")
