"""CPU model of the sampler's window rule (rambl_amd/csrc/sc_kernels.hip, urn_chain_q): a draw evaluated in fp32
with the counts in front of the window, p draws earlier, is final when no boundary above the target u*T lies within
eps*T + u*p of it and none below it within eps*T + (1-u)*p: each earlier draw adds L <= 1 to one count, which moves
cum_s - u*T up by at most (1-u) and down by at most u.  The model runs that rule with numpy fp32 arithmetic next to
the plain sequential fp64 chain (the reference's order of draws) and checks that every accepted decision is the
sequential one -- the property the device kernel relies on -- and that it accepts more draws per pass than the
symmetric rule (margin eps*T + p on both sides) it replaced."""
import numpy as np
import pytest


def sequential_chain(a0, L, u):
    """fp64, one draw at a time: first s with cum_s >= u*T, last strain forced (discrete_distribution)."""
    a = a0.astype(np.float64).copy()
    Q, S = L.shape
    out = np.empty(len(u), dtype=np.int64)
    for t, ut in enumerate(u):
        w = a * L[t % Q]
        cum = np.cumsum(w)
        c = int(np.searchsorted(cum[:-1], ut * cum[-1], side="left"))
        out[t] = c
        a[c] += 1.0
    return out


def windowed_chain(a0, L, u, width=64, one_sided=True):
    """The device rule in fp32; returns (decisions, passes, fallbacks)."""
    Q, S = L.shape
    Lf = L.astype(np.float32)
    a0f = a0.astype(np.float32)
    k = np.zeros(S, dtype=np.float64)                      # exact integer counts
    eps = np.float32((16 * ((S + 15) // 16) + 12) * 1.5e-7)
    out = np.empty(len(u), dtype=np.int64)
    t = passes = fallbacks = 0
    while t < len(u):
        passes += 1
        n = min(width, len(u) - t)
        af = (a0f + k.astype(np.float32)).astype(np.float32)
        adv = n
        cs = np.zeros(n, dtype=np.int64)
        for p in range(n):
            row = Lf[(t + p) % Q]
            cum = np.zeros(S, dtype=np.float32)
            run = np.float32(0)
            for s in range(S):                               # sequential fp32 FMA chain (rounded product + sum is a superset of its error)
                run = np.float32(run + np.float32(af[s] * row[s]))
                cum[s] = run
            T = cum[-1]
            tgt = np.float32(np.float32(u[t + p]) * T)
            d = (cum[:-1] - tgt).astype(np.float32)
            c = int(np.sum(d < 0))
            if one_sided:
                uf = np.float32(u[t + p])
                up = np.float32(np.min(d[d >= 0])) if np.any(d >= 0) else np.float32(np.inf)     # nearest boundary at / above the target
                dn = np.float32(np.min(-d[d < 0])) if np.any(d < 0) else np.float32(np.inf)      # nearest one below it
                pf = np.float32(p) + np.float32(1e-37)
                lim_up = np.float32(uf * pf + np.float32(eps * T))
                lim_dn = np.float32((np.float32(1) - uf) * pf + np.float32(eps * T))
                ok = (dn >= lim_dn) and (up >= lim_up)
            else:
                dmin = np.float32(np.min(np.abs(d))) if S > 1 else np.float32(1e30)
                ok = dmin >= np.float32(eps * T + np.float32(p) + np.float32(1e-37))
            if not ok:
                adv = p
                break
            cs[p] = c
        if adv == 0:                                         # within the fp32 bound of a boundary: exact evaluation
            fallbacks += 1
            a = a0.astype(np.float64) + k
            cum = np.cumsum(a * L[t % Q])
            cs[0] = int(np.searchsorted(cum[:-1], u[t] * cum[-1], side="left"))
            adv = 1
        for p in range(adv):
            out[t + p] = cs[p]
            k[cs[p]] += 1.0
        t += adv
    return out, passes, fallbacks


@pytest.mark.parametrize("seed", range(12))
def test_window_rule_reproduces_the_sequential_chain(seed):
    rng = np.random.default_rng(seed)
    S = int(rng.integers(2, 40))
    Q = int(rng.integers(3, 120))
    # weight rows as the sampler sees them: max 1 per row, a few strains near 1, the rest tiny or zero
    ll = rng.choice([0.0, -0.01, -5.0, -12.0, -40.0, -200.0], size=(Q, S), p=[0.25, 0.1, 0.25, 0.2, 0.15, 0.05])
    ll -= ll.max(axis=1, keepdims=True)
    L = np.exp(ll)
    a0 = rng.uniform(0.0, 60.0, size=S) * (rng.random(S) < 0.8)
    a0[rng.integers(0, S)] += 5.0
    n_draws = 1500
    u = rng.random(n_draws)
    # a few uniforms placed on top of boundaries of the first draws, and at the ends of the range
    u[::97] = np.clip(np.round(u[::97], 3), 0.0, 0.999999)
    u[5] = 0.0
    u[6] = 1.0 - 2.0 ** -53
    exp = sequential_chain(a0, L, u)
    got, passes, fallbacks = windowed_chain(a0, L, u)
    assert np.array_equal(got, exp)
    assert passes < n_draws                                  # the window does accept several draws per pass
    got128, passes128, _ = windowed_chain(a0, L, u, width=128)
    assert np.array_equal(got128, exp)
    old, passes_old, _ = windowed_chain(a0, L, u, one_sided=False)
    assert np.array_equal(old, exp) and passes <= passes_old    # never fewer draws per pass than the symmetric margin
