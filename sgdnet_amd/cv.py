"""cv_sgdnet(): k-fold cross-validation over alpha x lambda; host mirror of the reference's
R/cv_sgdnet.R:113-300.

The n_alpha * nfolds fold fits are independent calls of the same C-ABI entry point
(sgdnet_fit_sparse / sgdnet_fit_dense), so they fan out over the GPUs of a node with no
data-path collective: `devices=[0, 1, ...]` runs them from a thread pool, one fit per device at
a time (ctypes releases the GIL for the duration of the native call).

Two behaviours of the reference are kept because results depend on them, and flagged:
  * R/cv_sgdnet.R:182-183 trains on fold j (`train_ind <- j == foldid`) and scores on the other
    folds; `train_on="rest"` gives the conventional assignment instead.
  * R/cv_sgdnet.R:130 densifies x (`as.matrix`); here x stays sparse unless `densify=True`
    (a 10M x 10k matrix cannot be densified) -- same optimum, different standardisation path.
With one device and an `rng`, fits consume R's generator in the reference's order (full fits,
sample() for the fold ids, fold fits); with several devices every fit gets its own seed.
"""
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass

import numpy as np

from .api import sgdnet
from .score import _MEASURES, score
from .solver import RRng


@dataclass
class CvSgdnet:
    alpha: np.ndarray
    lambda_: list
    cv_summary: np.ndarray          # columns: alpha, lambda, mean, sd, ci_lo, ci_up
    cv_raw: list
    name: str
    fit: object
    alpha_min: float
    lambda_min: float
    lambda_1se: float
    foldid: np.ndarray


def r_cut(x, breaks):
    """as.numeric(cut(x, breaks)) for a single number of breaks (R's cut.default)."""
    x = np.asarray(x, dtype=np.float64)
    nb = int(breaks) + 1
    lo, hi = x.min(), x.max()
    dx = hi - lo
    if dx == 0:
        dx = abs(lo)
        b = np.linspace(lo - dx / 1000, hi + dx / 1000, nb)
    else:
        b = np.linspace(lo, hi, nb)
        b[0], b[-1] = lo - dx / 1000, hi + dx / 1000
    return np.searchsorted(b, x, side="left")                    # intervals (b_i, b_i+1], codes 1..breaks


def col_sd(m):
    """R/utils.R:38-46."""
    n = m.shape[0]
    var = np.mean(m ** 2, axis=0) - np.mean(m, axis=0) ** 2
    return np.sqrt(var * n / (n - 1))


def summarize_cv_raw(cv_raw):
    """R/cv_sgdnet.R:286-292: mean, sd, mean - sd, mean + sd per lambda."""
    bar, sd = cv_raw.mean(axis=0), col_sd(cv_raw)
    return np.column_stack([bar, sd, bar - sd, bar + sd])


def find_optimum(summary):
    """R/cv_sgdnet.R:262-276 for one alpha."""
    lam, mean, sd = summary[:, 1], summary[:, 2], summary[:, 3]
    i = int(np.argmin(mean))
    within = mean <= mean[i] + sd[i]
    return dict(alpha_min=summary[i, 0], lambda_min=lam[i], lambda_1se=lam[within].max(), error_min=mean[i])


def cv_sgdnet(x, y, alpha=1, lambda_=None, nfolds=10, foldid=None, type_measure="deviance", *, family="gaussian",
              devices=None, rng=None, seed=0, train_on="fold", densify=False, **fit_args):
    import scipy.sparse as sp

    alpha = np.atleast_1d(np.asarray(alpha, dtype=np.float64))
    if not (nfolds > 2 and alpha.size > 0):
        raise ValueError("nfolds > 2, is.numeric(alpha), length(alpha) > 0 are not all TRUE")
    if type_measure not in _MEASURES[family]:
        raise ValueError("'arg' should be one of " + ", ".join(f"'{m}'" for m in _MEASURES[family]))
    if densify and sp.issparse(x):
        x = np.asarray(x.todense())
    if sp.issparse(x):
        x = sp.csr_matrix(x)
    else:
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
    y = np.asarray(y)
    n = x.shape[0]
    if nfolds > n:
        raise ValueError("you cannot have more folds than samples.")
    if isinstance(lambda_, list) and lambda_ and isinstance(lambda_[0], (list, tuple, np.ndarray)):
        if len(lambda_) != alpha.size:
            raise ValueError("the length of the lambda list needs to match the number of alpha.")
        lam_in = [np.asarray(l, dtype=np.float64) for l in lambda_]
    elif lambda_ is not None:
        if alpha.size > 1:
            raise ValueError("you need a list of lambdas (or have it set at NULL) when you have multiple alphas.")
        lam_in = [np.asarray(lambda_, dtype=np.float64)]
    else:
        lam_in = [None] * alpha.size

    devices = [0] if not devices else list(devices)
    sequential = len(devices) == 1
    if rng is None:
        rng = RRng(seed)

    def one_fit(xx, yy, lam, a, dev, fit_seed):
        kw = dict(fit_args)
        if sequential:
            kw["rng"] = rng                       # R's global generator, advanced by every fit
        else:
            kw["seed"] = fit_seed
        return sgdnet(xx, yy, family=family, alpha=float(a), lambda_=lam, device=dev, **kw)

    fits = [one_fit(x, y, lam_in[i], alpha[i], devices[0], seed + 1 + i) for i in range(alpha.size)]
    lam = [f.lambda_ for f in fits]

    if foldid is None:
        foldid = r_cut(rng.sample(n), nfolds)      # as.numeric(cut(sample(n_samples), nfolds))
    else:
        foldid = np.asarray(foldid)
        if foldid.size != n:
            raise ValueError("the length of `foldid` must match the number of samples")
        nfolds = np.unique(foldid).size
    fold_values = np.arange(1, nfolds + 1) if np.all(np.isin(foldid, np.arange(1, nfolds + 1))) else np.unique(foldid)

    def fold_job(job):
        i, j, worker, fit_seed = job
        dev = devices[worker]
        sel = foldid == fold_values[j]
        train = sel if train_on == "fold" else ~sel
        test = ~train
        fit = one_fit(x[train], y[train], lam[i], alpha[i], dev, fit_seed)
        # R/score.R auc(): stats::runif(2 n) per lambda, in column order, from the global generator -- the draws order
        # equal probabilities AND move the stream the next fold's fit starts from; drawn on the device (sgdnet_auc_*_rng)
        tie_rng = rng if (type_measure == "auc" and sequential) else None
        return i, j, score(fit, x[test], y[test], type_measure, device=dev, rng=tie_rng)

    jobs = [(i, j, (i * nfolds + j) % len(devices), seed + 1000 + i * nfolds + j)
            for i in range(alpha.size) for j in range(nfolds)]               # (alpha, fold, worker, seed)
    cv_raw = [np.full((nfolds, lam[i].size), np.nan) for i in range(alpha.size)]
    if sequential:
        results = map(fold_job, jobs)
    else:
        # one worker per entry of `devices` (list a device twice to run two fits on it at a time)
        pools = [ThreadPoolExecutor(max_workers=1) for _ in devices]
        futures = [pools[job[2]].submit(fold_job, job) for job in jobs]
        results = (f.result() for f in futures)
    for i, j, sc in results:
        cv_raw[i][j, :] = sc
    if not sequential:
        for pool in pools:
            pool.shutdown()

    blocks = []
    for i in range(alpha.size):
        blocks.append(np.column_stack([np.full(lam[i].size, alpha[i]), lam[i], summarize_cv_raw(cv_raw[i])]))
    summary = np.vstack(blocks)
    optima = [find_optimum(b) for b in blocks]
    best = int(np.argmin([o["error_min"] for o in optima]))
    if type_measure == "deviance":
        name = {"gaussian": "Mean-Squared Error", "mgaussian": "Mean-Squared Error",
                "binomial": "Binomial Deviance", "multinomial": "Multnomial Deviance"}[family]
    else:
        name = {"mse": "Mean-Squared Error", "mae": "Mean Absolute Error", "class": "Misclassification Error",
                "auc": "AUC"}[type_measure]
    return CvSgdnet(alpha=alpha, lambda_=lam, cv_summary=summary, cv_raw=cv_raw, name=name, fit=fits[best],
                    alpha_min=optima[best]["alpha_min"], lambda_min=optima[best]["lambda_min"],
                    lambda_1se=optima[best]["lambda_1se"], foldid=foldid)
