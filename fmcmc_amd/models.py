"""Closed-form log-posterior families selected through MCMC()'s `fun` argument.

fmcmc calls an arbitrary R closure `fun(theta, ...)` every iteration (R/mcmc.R:683-718,754); a
fused GPU kernel cannot, so the engine ships the families the reference's own documentation uses:

  gaussian_linreg(X, y)   README.md:128-139 (guarded) / :356-361 (guard=False)
  logistic(X, y)          vignettes/workflow-with-fmcmc.Rmd:35-41 (prior -sum(beta^2)/8)
  iid_normal(D)           R/mcmc.R:141-144

Each constructor returns a LogPosterior: a tagged object carrying the data (moved to HBM once per
device) that MCMC() recognises.  Calling it, fun(theta), evaluates the same closed form in numpy for
inspection only -- MCMC() never calls it.
"""
import numpy as np

from . import _abi as abi


class LogPosterior:
    def __init__(self, family, X, y, intercept, guard, prior_div, names=None):
        self.family = family
        self.y = np.ascontiguousarray(np.asarray(y, dtype=np.float64))
        if X is not None:
            X = np.asarray(X, dtype=np.float64)
            if X.ndim == 1:
                X = X[:, None]
            if X.shape[0] != self.y.shape[0]:
                raise ValueError("X has %d rows but y has %d" % (X.shape[0], self.y.shape[0]))
        self.X = X
        self.p = 0 if X is None else X.shape[1]
        self.intercept, self.guard, self.prior_div = bool(intercept), bool(guard), float(prior_div)
        self.names = names
        self._dev = {}

    @property
    def k(self):
        if self.family == abi.FAM_GAUSSIAN_LINREG:
            return int(self.intercept) + self.p + 1
        if self.family == abi.FAM_LOGISTIC:
            return int(self.intercept) + self.p
        return 2

    def device_model(self, device):
        from .engine import DeviceModel
        key = str(device)
        if key not in self._dev:
            self._dev[key] = DeviceModel(self.family, self.X, self.y, self.intercept, self.guard,
                                         self.prior_div, device=device)
        return self._dev[key]

    def __call__(self, theta):
        th = np.asarray(theta, dtype=np.float64)
        ic = int(self.intercept)
        with np.errstate(all="ignore"):
            if self.family == abi.FAM_LOGISTIC:
                eta = (th[0] if ic else 0.0) + (self.X @ th[ic:] if self.p else 0.0)
                s = np.where(self.y != 0, eta, -eta)
                f = float(np.sum(np.where(s < 0, s - np.log1p(np.exp(s)), -np.log1p(np.exp(-s)))))
                if self.prior_div:
                    f -= float(np.sum(th ** 2)) / self.prior_div
            else:
                if self.family == abi.FAM_IID_NORMAL:
                    mu, sigma = th[0], th[1]
                else:
                    mu = (th[0] if ic else 0.0) + (self.X @ th[ic:ic + self.p] if self.p else 0.0)
                    sigma = th[ic + self.p]
                if sigma < 0:
                    f = float("nan")
                elif sigma == 0:
                    f = float("-inf")
                else:
                    f = float(-self.y.size * (np.log(sigma) + 0.9189385332046727) -
                              0.5 * np.sum((self.y - mu) ** 2) / sigma ** 2)
        if self.guard and not np.isfinite(f):
            return float("-inf")
        return f


def gaussian_linreg(X, y, intercept=True, guard=True):
    """sum(dnorm(y - (b0 + X b), sd = sigma, log = TRUE)); theta = (b0, b_1..b_p, sigma)."""
    return LogPosterior(abi.FAM_GAUSSIAN_LINREG, X, y, intercept, guard, 0.0)


def logistic(X, y, intercept=False, prior_div=8.0, guard=False):
    """Bernoulli-logit log-likelihood with the N(0, prior_div/2) prior of the vignette.
    Pass the model matrix as the vignette does (own intercept column) or set intercept=True."""
    return LogPosterior(abi.FAM_LOGISTIC, X, y, intercept, guard, prior_div)


def iid_normal(D, guard=False):
    """sum(log(dnorm(D, mu, sigma))); theta = (mu, sigma)."""
    return LogPosterior(abi.FAM_IID_NORMAL, None, D, True, guard, 0.0)
