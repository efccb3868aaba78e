"""Host utilities fmcmc exports next to its kernels (not on the device hot path, where the same formulas run inside
the owner wavefronts):

  mean_recursive(X_t, Mean_t_prev, t.)                          R/recursive.R:124-139
  cov_recursive(X_t, Cov_t, Mean_t_prev, t., Mean_t, eps, Sd, Ik)  R/recursive.R:63-120
  reflect_on_boundaries(x, lb, ub, which)                       R/kernel.R:450-493
"""
import numpy as np


def mean_recursive(X_t, Mean_t_prev, t_):
    """(Mean_{t-1} * t + X_t) / (t + 1); a matrix X_t is folded in row by row (R/recursive.R:129-136)."""
    X_t = np.asarray(X_t, dtype=np.float64)
    m = np.asarray(Mean_t_prev, dtype=np.float64).reshape(-1)
    if X_t.ndim == 1:
        return (m * t_ + X_t) / (t_ + 1)
    out = np.empty_like(X_t)
    for i in range(X_t.shape[0]):
        prev = m if i == 0 else out[i - 1]
        out[i] = (prev * (t_ + i) + X_t[i]) / (t_ + i + 1)
    return out


def cov_recursive(X_t, Cov_t, Mean_t_prev, t_, Mean_t=None, eps=0.0, Sd=1.0, Ik=None):
    """(t-1)/t Cov + Sd/t (t m_ m_' - (t+1) m m' + x x' + eps Ik) (R/recursive.R:112-118).
    Matrix X_t: returns the stack [nrow][k][k] of successive covariances (R returns k x k x nrow)."""
    X_t = np.asarray(X_t, dtype=np.float64)
    Cov_t = np.asarray(Cov_t, dtype=np.float64)
    k = Cov_t.shape[0]
    if Ik is None:
        Ik = np.eye(k)
    if Mean_t is None:
        Mean_t = mean_recursive(X_t, Mean_t_prev, t_)
    Mean_t = np.asarray(Mean_t, dtype=np.float64)
    mp = np.asarray(Mean_t_prev, dtype=np.float64).reshape(-1)
    if X_t.ndim == 2:
        out = np.empty((X_t.shape[0], k, k))
        for i in range(X_t.shape[0]):
            out[i] = cov_recursive(X_t[i], Cov_t if i == 0 else out[i - 1], mp if i == 0 else Mean_t[i - 1], t_ + i,
                                   Mean_t=Mean_t[i], eps=eps, Sd=Sd, Ik=Ik)
        return out
    return (t_ - 1) / t_ * Cov_t + Sd / t_ * (t_ * np.outer(mp, mp) - (t_ + 1) * np.outer(Mean_t, Mean_t)
                                              + np.outer(X_t, X_t) + eps * Ik)


def reflect_on_boundaries(x, lb, ub, which):
    """Closed-form multi-fold reflection of x[which] into [lb, ub] (R/kernel.R:450-493). `which`: 0-based indices."""
    x = np.array(x, dtype=np.float64, copy=True)
    lb = np.asarray(lb, dtype=np.float64)
    ub = np.asarray(ub, dtype=np.float64)
    which = np.asarray(which, dtype=np.int64)
    d = ub - lb
    above = which[x[which] > ub[which]]
    below = which[x[which] < lb[which]]
    if above.size:
        e = x[above] - ub[above]
        odd = np.floor_divide(e, d[above]) % 2
        e = np.mod(e, d[above])
        x[above] = (lb[above] + e) * odd + (ub[above] - e) * (1 - odd)
    if below.size:
        e = lb[below] - x[below]
        odd = np.floor_divide(e, d[below]) % 2
        e = np.mod(e, d[below])
        x[below] = (ub[below] - e) * odd + (lb[below] + e) * (1 - odd)
    return x
