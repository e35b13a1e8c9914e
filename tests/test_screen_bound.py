"""CPU: the two inequalities the screen of fvconvert's shape 3 rests on (csrc/gmmmap_screen.hpp), checked numerically with
numpy emulations -- no GPU, no library call:
  (1) a partial sum of the eigen-expansion of (x - mu)' inv(S) (x - mu) over the largest eigenpairs of inv(S) is a lower
      bound of it (so lc - bound / 2 is an upper bound of the mixture's log-density);
  (2) the bf16-split evaluation  a^ = Ph xh + Ph xl + Pl xh - c  with FP32 accumulation differs from a = P x - c by at most
      eps = 2^-12 (|P| |x| + |c|) -- the margin the kernel subtracts from |a^| before squaring -- also under heavy
      cancellation (c thousands of times a) and for any summation order."""
import numpy as np


def bf16_round(x32):
    """float32 -> nearest bf16 (ties to even), returned as float32 (the kernel's split_bf16)."""
    u = np.asarray(x32, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return r.astype(np.uint32).view(np.float32)


def split(x64):
    xf = np.asarray(x64, dtype=np.float64).astype(np.float32)
    hi = bf16_round(xf)
    lo = bf16_round((xf - hi).astype(np.float32))        # xf - hi is exact in float32
    return hi, lo


def test_partial_eigen_sum_is_a_lower_bound():
    rng = np.random.default_rng(0)
    for D in (16, 24, 40):
        for _ in range(20):
            Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
            lam = np.exp(rng.uniform(np.log(1e-5), 0.0, D))
            S = (Q * lam) @ Q.T
            S = 0.5 * (S + S.T)
            Sinv = np.linalg.inv(S)
            kap, V = np.linalg.eigh(Sinv)
            order = np.argsort(kap)[::-1]
            P = (np.sqrt(kap[order[:4]])[:, None] * V[:, order[:4]].T)          # rows sqrt(kappa_i) v_i'
            mu = rng.standard_normal(D)
            X = mu + rng.standard_normal((200, D)) * rng.uniform(0.01, 3.0, (200, 1))
            full = np.einsum("td,de,te->t", X - mu, Sinv, X - mu)
            for rows in (1, 2, 4):
                part = (((X - mu) @ P[:rows].T) ** 2).sum(1)
                assert np.all(part <= full * (1 + 1e-9) + 1e-9)


def test_bf16_split_error_margin_holds():
    rng = np.random.default_rng(1)
    worst = 0.0
    for D in (16, 28, 40):
        for scale_p, shift in ((1.0, 0.0), (300.0, 0.0), (300.0, 500.0), (3.0, -2000.0), (1e3, 50.0)):
            P = rng.standard_normal((64, D)) * scale_p * np.exp(rng.uniform(-3, 0, (64, 1)))
            mu = rng.standard_normal(D) + shift
            X = mu + rng.standard_normal((256, D)) * rng.uniform(1e-3, 2.0, (256, 1))
            c = P @ mu
            a = X @ P.T - c                                                       # (T, rows), float64: the quantity bounded
            ph, pl = split(P)
            xh, xl = split(X)
            # the products of two bf16 numbers are exact in float32; accumulate the 3 D terms in float32, in two different orders
            terms = np.concatenate([xh[:, None, :] * ph[None], xl[:, None, :] * ph[None], xh[:, None, :] * pl[None]], axis=2).astype(np.float32)
            for order in (slice(None), slice(None, None, -1)):
                acc = (-c).astype(np.float32)[None, :] * np.ones((X.shape[0], 1), np.float32)
                for k in range(terms.shape[2])[order]:
                    acc = (acc + terms[:, :, k]).astype(np.float32)
                eps = 2.0 ** -12 * (np.linalg.norm(P, axis=1)[None, :] * np.linalg.norm(X, axis=1)[:, None] + np.abs(c)[None, :])
                err = np.abs(acc.astype(np.float64) - a)
                assert np.all(err <= eps), float((err / eps).max())
                worst = max(worst, float((err / eps).max()))
                # ... hence the certified lower bound of a^2 never exceeds a^2
                lb = np.maximum(np.abs(acc.astype(np.float64)) - eps, 0.0) ** 2
                assert np.all(lb <= a * a * (1 + 1e-12) + 1e-300)
    assert worst < 0.6          # the margin is about twice what the arithmetic needs (sequential FP32 accumulation, the worst order)
