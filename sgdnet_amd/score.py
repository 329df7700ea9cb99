"""score(): prediction error along the lambda path; host mirror of the reference's R/score.R."""
import numpy as np

from .predict import predict

_MEASURES = {"gaussian": ("deviance", "mse", "mae"), "mgaussian": ("deviance", "mse", "mae"),
             "binomial": ("deviance", "mse", "mae", "class", "auc"),
             "multinomial": ("deviance", "mse", "mae", "class")}
PROB_MIN = 1e-05                                                  # R/score.R:88, 133


def auc(y01, prob, weights=None, tie_break=None):
    """R/score.R:220-250.  y01: (n, 2) indicator matrix or a 0/1 vector.  The reference breaks
    ties between equal probabilities with stats::runif(); pass tie_break (same length as the
    stacked vector) to reproduce a given draw, default: stable order."""
    y01 = np.asarray(y01, dtype=np.float64)
    if y01.ndim == 2:
        ny = y01.shape[0]
        w = np.ones(ny) if weights is None else np.asarray(weights, dtype=np.float64)
        return auc(np.r_[np.zeros(ny), np.ones(ny)], np.r_[prob, prob], (w[:, None] * y01).ravel("F"), tie_break)
    prob = np.asarray(prob, dtype=np.float64)
    if weights is None:
        from scipy.stats import rankdata
        r = rankdata(prob)
        n1 = y01.sum()
        n0 = y01.size - n1
        u = r[y01 == 1].sum() - n1 * (n1 + 1) / 2
        return float(np.exp(np.log(u) - np.log(n1) - np.log(n0)))
    tb = np.arange(prob.size) if tie_break is None else np.asarray(tie_break)
    op = np.lexsort((tb, prob))
    y, w = y01[op], np.asarray(weights, dtype=np.float64)[op]
    cw = np.cumsum(w)
    w1 = w[y == 1]
    cw1 = np.cumsum(w1)
    wauc = np.log(np.sum(w1 * (cw[y == 1] - cw1)))
    sumw1 = cw1[-1]
    sumw2 = cw[-1] - sumw1
    return float(np.exp(wauc - np.log(sumw1) - np.log(sumw2)))


def score(fit, x, y, type_measure="deviance", s=None):
    """score.sgdnet_<family>: one value per lambda (or per entry of s)."""
    fam = fit.family
    if type_measure not in _MEASURES[fam]:
        raise ValueError("'arg' should be one of " + ", ".join(f"'{m}'" for m in _MEASURES[fam]))
    s = fit.lambda_ if s is None else s
    y = np.asarray(y)
    if fam == "gaussian":                                         # R/score.R:55-70
        yh = predict(fit, x, s)
        d = yh - y.reshape(-1, 1)
        return np.mean(np.abs(d), axis=0) if type_measure == "mae" else np.mean(d ** 2, axis=0)
    if fam == "mgaussian":                                        # :172-186 (colSums over samples, mean over responses)
        yh = predict(fit, x, s)
        d = yh - np.asarray(y, dtype=np.float64)[:, :, None]
        e = np.abs(d) if type_measure == "mae" else d ** 2
        return e.sum(axis=0).mean(axis=0)
    levels = np.unique(y)
    Y = (y.reshape(-1, 1) == levels.reshape(1, -1)).astype(np.float64)     # diag(K)[as.numeric(y), ]
    if fam == "binomial":                                         # :74-115
        ph = predict(fit, x, s, type="response")
        y1, y2 = Y[:, 0:1], Y[:, 1:2]
        if type_measure == "auc":
            return np.array([auc(Y, ph[:, i]) for i in range(ph.shape[1])])
        if type_measure == "mse":
            return np.mean((ph + y1 - 1) ** 2 + (ph - y2) ** 2, axis=0)
        if type_measure == "mae":
            return np.mean(np.abs(ph + y1 - 1) + np.abs(ph - y2), axis=0)
        if type_measure == "class":
            return np.mean(y1 * (ph > 0.5) + y2 * (ph <= 0.5), axis=0)
        ph = np.clip(ph, PROB_MIN, 1 - PROB_MIN)
        lp = y1 * np.log(1 - ph) + y2 * np.log(ph)
        return np.mean(2 * (0.0 - lp), axis=0)                    # ly = sum y log y = 0 for 0/1 responses
    ph = predict(fit, x, s, type="response")                      # multinomial, :119-168: (n, K, L)
    Y3 = Y[:, :, None]
    if type_measure == "mse":
        return ((Y3 - ph) ** 2).sum(axis=1).mean(axis=0)
    if type_measure == "mae":
        return np.abs(Y3 - ph).sum(axis=1).mean(axis=0)
    if type_measure == "class":
        cls = np.argmax(ph, axis=1)                                # (n, L)
        return 1.0 - np.take_along_axis(Y, cls, axis=1).mean(axis=0)
    ph = np.clip(ph, PROB_MIN, 1 - PROB_MIN)
    return (2 * (0.0 - Y3 * np.log(ph))).sum(axis=1).mean(axis=0)
