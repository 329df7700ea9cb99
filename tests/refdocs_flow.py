"""The reference's documented examples (man/*.Rd \\examples, printed in docs/reference/*.html),
replayed against a pluggable backend (TEST INFRASTRUCTURE).

pkgdown 1.1.0 runs all examples in one R session after one set.seed(1014), topic by topic in
alphabetical order, so each example starts from the generator state the previous one left
unless it calls set.seed() itself.  The chains below follow that order; every native fit draws
n_samples uniforms per epoch (src/saga-dense.h:152, src/saga-sparse.h:261), so a chain only
reproduces the printed numbers if every fit before it ran exactly the reference's number of
epochs.  Expected values: tests/golden/refdocs.npz (tests/golden/make_refdocs_fixtures.py).

Backends:
  OracleBackend -- the CPU oracle behind the reference's front-end logic (CPU tests)
  HipBackend    -- sgdnet_amd.sgdnet()/cv_sgdnet() = libsgdnet_hip.so in exact mode (GPU tests)
"""
import contextlib
import ctypes
import os

import numpy as np
import scipy.sparse as sp

import rcompat as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    d = dict(np.load(os.path.join(GOLDEN, "refdocs.npz")))
    dim = tuple(int(v) for v in d["heart_dim"])
    d["HX"] = sp.csc_matrix((d["heart_x"], d["heart_i"], d["heart_p"]), shape=dim).tocsr()
    d["HY"] = d["heart_levels"][d["heart_y"] - 1]
    d["WX"] = d["wine_x"]
    d["WY"] = d["wine_levels"][d["wine_y"] - 1]
    ir = np.load(os.path.join(GOLDEN, "iris.npz"))
    d["IX"] = ir["x"]
    d["IY"] = np.array(["setosa", "versicolor", "virginica"])[ir["y"].astype(int)]
    ab = np.load(os.path.join(GOLDEN, "abalone.npz"))
    d["AX"], d["AY"] = ab["x"], ab["y"]
    return d


class OracleBackend:
    name = "oracle"

    def rng(self, seed):
        from oracle import pyoracle as po

        class _Rng(po.Rng):
            def sample(self, n, size=None):          # R 3.5: sample.kind = "Rounding"
                return R.sample_rounding(self, n, size)
        return _Rng(seed)

    def clone(self, rng):
        c = self.rng(0)
        ctypes.memmove(ctypes.byref(c.state), ctypes.byref(rng.state), ctypes.sizeof(rng.state))
        return c

    def fit(self, x, y, rng, **kw):
        return R.oracle_sgdnet(x, y, rng=rng, **kw)

    def cv(self, x, y, rng, **kw):
        import importlib
        CV = importlib.import_module("sgdnet_amd.cv")
        S = importlib.import_module("sgdnet_amd.score")
        with _patched(CV, sgdnet=R.oracle_sgdnet, score=lambda fit, xx, yy, m, device=None, tie_break=None, rng=None: S.score(fit, xx, yy, m, tie_break=tie_break, rng=rng)):
            return CV.cv_sgdnet(x, y, rng=rng, densify=True, **kw)


class HipBackend:
    name = "hip"

    def __init__(self, mode="exact"):
        self.mode = mode

    def rng(self, seed):
        import sgdnet_amd as sa
        return sa.RRng(seed, sample_kind="Rounding")

    def clone(self, rng):
        c = self.rng(0)
        ctypes.memmove(ctypes.byref(c.state), ctypes.byref(rng.state), ctypes.sizeof(rng.state))
        return c

    def fit(self, x, y, rng, **kw):
        import sgdnet_amd as sa
        return sa.sgdnet(x, y, rng=rng, mode=self.mode, **kw)

    def cv(self, x, y, rng, **kw):
        import sgdnet_amd as sa
        return sa.cv_sgdnet(x, y, rng=rng, densify=True, mode=self.mode, **kw)


@contextlib.contextmanager
def _patched(module, **names):
    old = {k: getattr(module, k) for k in names}
    try:
        for k, v in names.items():
            setattr(module, k, v)
        yield
    finally:
        for k, v in old.items():
            setattr(module, k, v)


def _holdout(n, train_1based):
    mask = np.ones(n, dtype=bool)
    mask[train_1based - 1] = False
    return mask


# ------------------------------------------------------------------ chain A: set.seed(1014)
def example_coef(be, d):
    """man/coef.sgdnet.Rd: the first topic with examples, so it starts at set.seed(1014).
    Returns the 3 x 100 coefficient table (intercept, V1, V2)."""
    from sgdnet_amd.predict import coef
    rng = be.rng(1014)
    x = R.rnorm(rng, 100).reshape(50, 2, order="F")
    y = R.rnorm(rng, 50)
    return coef(be.fit(x, y, rng))


# ------------------------------------------------------------------ chain B: cv_sgdnet.Rd -> deviance.sgdnet.Rd
def example_cv_heart_then_deviance(be, d):
    """man/cv_sgdnet.Rd (set.seed(1)) followed by man/deviance.sgdnet.Rd (inherits the state)."""
    from sgdnet_amd.predict import predict
    rng = be.rng(1)
    n = 270
    tr = rng.sample(n, int(np.floor(0.8 * n)))
    cvf = be.cv(d["HX"][tr - 1], d["HY"][tr - 1], rng, family="binomial", nfolds=7, alpha=[0, 1])
    link = predict(cvf.fit, d["HX"][_holdout(n, tr)], s=cvf.lambda_min)[:, 0]
    fit = be.fit(d["WX"], d["WY"], rng, family="multinomial")
    return dict(link=link, alpha_min=cvf.alpha_min, lambda_min=cvf.lambda_min,
                deviance=(1.0 - fit.dev_ratio) * fit.nulldev, nulldev=fit.nulldev)


# ------------------------------------------------------------------ chain C: predict.cv_sgdnet.Rd -> predict.sgdnet.Rd
#                                                                     -> print.cv_sgdnet.Rd -> print.sgdnet.Rd
def example_predict_chain(be, d, student_lambda0_nudge=0.0):
    """student_lambda0_nudge: relative increase of the first (= lambda_max) entry of the student
    path.  At lambda_max the group-lasso threshold test of src/penalties.h:72-77 is decided by the
    last bits of LambdaMax (an Eigen GEMM in the reference): see DESIGN.md 2."""
    from sgdnet_amd.predict import predict
    out = {}
    rng = be.rng(1)
    tr = rng.sample(150, 100)
    cvf = be.cv(d["IX"][tr - 1], d["IY"][tr - 1], rng, family="multinomial", nfolds=5)
    out["iris_class"] = predict(cvf.fit, d["IX"][_holdout(150, tr)], s=cvf.lambda_min, type="class")[:, 0]
    # predict.sgdnet.Rd
    n = 4177
    tr = rng.sample(n, int(np.floor(0.8 * n)))
    be.fit(d["AX"][tr - 1], d["AY"][tr - 1], rng)
    n = 270
    tr = rng.sample(n, int(np.floor(0.8 * n)))
    fb = be.fit(d["HX"][tr - 1], d["HY"][tr - 1], rng, family="binomial")          # sparse x, standardize = TRUE
    out["heart_class"] = predict(fb, d["HX"][_holdout(n, tr)], type="class", s=1.0 / n)[:, 0]
    n = 178
    tr = rng.sample(n, int(np.floor(0.8 * n)))
    be.fit(d["WX"][tr - 1], d["WY"][tr - 1], rng, family="multinomial", alpha=0.25)
    # (predict(..., exact = TRUE) fails in the docs before any refit: "object 'train_ind' not found")
    if student_lambda0_nudge:
        lam = be.fit(d["student_x"], d["student_y"], be.clone(rng), family="mgaussian").lambda_.copy()
        lam[0] *= 1.0 + student_lambda0_nudge
        fs = be.fit(d["student_x"], d["student_y"], rng, family="mgaussian", lambda_=lam)
    else:
        fs = be.fit(d["student_x"], d["student_y"], rng, family="mgaussian")
    out["student_nonzero"] = np.abs(fs.beta[0]).T > 0                              # (n_lambda, p)
    out["student_npasses"] = fs.npasses
    # print.cv_sgdnet.Rd
    disp, hp, drat = d["mtcars_disp_hp_drat"].T
    cvm = be.cv(drat.reshape(-1, 1), hp, rng, family="gaussian")
    sm = cvm.cv_summary
    i = int(np.flatnonzero(sm[:, 1] == cvm.lambda_min)[0])
    j = int(np.flatnonzero(sm[:, 1] == cvm.lambda_1se)[0])
    out["print_cv"] = np.vstack([sm[i], sm[j]])
    # print.sgdnet.Rd
    fp = be.fit(np.column_stack([drat, hp]), disp, rng)
    out["mtcars_lambda"], out["mtcars_df"], out["mtcars_dev"] = fp.lambda_, fp.df, fp.dev_ratio
    return out


# ------------------------------------------------------------------ chain D: score.Rd
def example_score_wine(be, d):
    from sgdnet_amd.score import score
    rng = be.rng(1)
    n = 178
    tr = rng.sample(n, int(np.floor(0.8 * n)))
    cvf = be.cv(d["WX"][tr - 1], d["WY"][tr - 1], rng, family="multinomial", nfolds=5, alpha=[0.5, 1])
    hold = _holdout(n, tr)
    return float(score(cvf.fit, d["WX"][hold], d["WY"][hold], "deviance", s=cvf.lambda_1se)[0])


def signif_round(x, digits):
    """What print(x, digits) shows of a number: `digits` significant digits."""
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.where(x == 0, 0, np.floor(np.log10(np.abs(x))))
    f = 10.0 ** (digits - 1 - e)
    return np.round(x * f) / f
