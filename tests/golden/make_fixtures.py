"""Generates the committed fixtures under tests/golden/ (run in the dev container only).

Inputs are DATA: the reference's bundled data sets (data/*.rda, RDX2/XDR + bzip2;
decoded by the minimal reader below) and sklearn's copy of R's iris table.  Expected
outputs come from the CPU oracle (oracle/pyoracle.py), whose parity status is stated in
oracle/sgdnet_oracle.h.  No reference source text is read or stored.

    python tests/golden/make_fixtures.py
"""
import bz2
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_DATA = "/root/reference/data"


class _RdaReader:
    """Just enough of R's XDR serialisation (version 2) for data frames, matrices,
    factors and dgCMatrix objects."""

    def __init__(self, raw):
        assert raw[:7] == b"RDX2\nX\n", raw[:8]
        self.b, self.i = raw, 7
        self.refs = []

    def u32(self):
        v = struct.unpack_from(">i", self.b, self.i)[0]
        self.i += 4
        return v

    def read_all(self):
        self.u32(); self.u32(); self.u32()          # format version, R versions
        return self.item()

    def item(self):
        flags = self.u32()
        t = flags & 0xFF
        has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
        if t == 254:                                 # NILVALUE
            return None
        if t == 253 or t == 242:                     # global env / empty env
            return None
        if t == 255:                                 # REFSXP
            return self.refs[(flags >> 8) - 1]
        if t == 1:                                   # SYMSXP
            s = self.item()
            self.refs.append(s)
            return s
        if t == 9:                                   # CHARSXP
            n = self.u32()
            if n == -1:
                return None
            s = self.b[self.i:self.i + n].decode("latin1")
            self.i += n
            return s
        if t in (2, 6):                              # LISTSXP / LANGSXP -> dict by tag
            out = {}
            k = 0
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else f"_{k}"
                out[tag] = self.item()
                k += 1
                flags = self.u32()
                t2 = flags & 0xFF
                if t2 == 254:
                    return out
                assert t2 in (2, 6), t2
                has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
        if t in (10, 13):                            # LGLSXP / INTSXP
            n = self.u32()
            v = np.frombuffer(self.b, dtype=">i4", count=n, offset=self.i).astype(np.int64)
            self.i += 4 * n
        elif t == 14:                                # REALSXP
            n = self.u32()
            v = np.frombuffer(self.b, dtype=">f8", count=n, offset=self.i).astype(np.float64)
            self.i += 8 * n
        elif t == 16:                                # STRSXP
            n = self.u32()
            v = [self.item() for _ in range(n)]
        elif t == 19:                                # VECSXP
            n = self.u32()
            v = [self.item() for _ in range(n)]
        elif t == 25:                                # S4SXP: slots are the attributes
            v = "S4"
        else:
            raise NotImplementedError(f"SEXP type {t} at {self.i}")
        attrs = self.item() if has_attr else {}
        return {"value": v, "attr": attrs or {}}


def load_rda(path):
    return _RdaReader(bz2.decompress(open(path, "rb").read())).read_all()


def _frame_to_matrix(obj):
    cols = [np.asarray(c["value"], dtype=np.float64) for c in obj["value"]]
    names = obj["attr"]["names"]["value"]
    return np.column_stack(cols), names


def abalone():
    top = load_rda(os.path.join(REF_DATA, "abalone.rda"))["abalone"]
    parts = dict(zip(top["attr"]["names"]["value"], top["value"]))
    x, names = _frame_to_matrix(parts["x"])
    y = np.asarray(parts["y"]["value"], dtype=np.float64)
    return x, y, names


def abalone_libsvm_layout(x9):
    """4177 x 9 in-repo frame -> the 4177 x 8 libsvm table it was made from."""
    return np.column_stack([1.0 + x9[:, 0] + 2.0 * x9[:, 1], x9[:, 2:]])


def main():
    from oracle import pyoracle as po
    from sklearn.datasets import load_iris

    # ---- C2: abalone 4177 x 9 (in-repo frame; the libsvm 8-column form merges the two
    #      sex dummies), gaussian, default path ----
    x, y, names = abalone()
    assert x.shape == (4177, 9) and y.shape == (4177,), (x.shape, y.shape)
    np.savez_compressed(os.path.join(HERE, "abalone.npz"), x=x, y=y, names=np.array(names))
    fit = po.fit(x, y, family="gaussian", alpha=1.0, nlambda=100, thresh=1e-3, maxit=1000, seed=2)
    np.savez_compressed(os.path.join(HERE, "abalone_gaussian_path.npz"),
                        a0=fit["a0"], beta=fit["beta"], lambda_=fit["lambda"],
                        dev_ratio=fit["dev_ratio"], npasses=fit["npasses"], nulldev=fit["nulldev"],
                        args=np.array("family=gaussian alpha=1 nlambda=100 thresh=1e-3 maxit=1000 "
                                      "standardize=TRUE intercept=TRUE seed=2 mode=exact"))

    # ---- C2 as BASELINE.json words it, "abalone 4177 x 8": the libsvm layout the reference's benchmark
    #      vignette reads (vignettes/benchmarks.Rmd:66).  The in-repo frame is that table with the 3-level sex
    #      code V1 expanded into two dummies (data-raw/datasets.R:13-25: sex = [V1 == 2], infant = [V1 == 3]),
    #      so V1 = 1 + sex + 2 infant and the other seven columns are unchanged. ----
    x8 = abalone_libsvm_layout(x)
    assert x8.shape == (4177, 8) and set(np.unique(x8[:, 0])) == {1.0, 2.0, 3.0}
    fit = po.fit(x8, y, family="gaussian", alpha=1.0, nlambda=100, thresh=1e-3, maxit=1000, seed=2)
    np.savez_compressed(os.path.join(HERE, "abalone8_gaussian_path.npz"),
                        a0=fit["a0"], beta=fit["beta"], lambda_=fit["lambda"],
                        dev_ratio=fit["dev_ratio"], npasses=fit["npasses"], nulldev=fit["nulldev"],
                        args=np.array("x = [1 + sex + 2 infant, length ... weight_shell] of abalone.npz; "
                                      "family=gaussian alpha=1 nlambda=100 thresh=1e-3 maxit=1000 "
                                      "standardize=TRUE intercept=TRUE seed=2 mode=exact"))

    # ---- C1: iris 150 x 4, multinomial, alpha = 0.8, default path ----
    ir = load_iris()
    xi, yi = ir.data, ir.target.astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "iris.npz"), x=xi, y=yi)
    fit = po.fit(xi, yi, family="multinomial", alpha=0.8, nlambda=100, thresh=1e-3, maxit=1000,
                 seed=1)
    np.savez_compressed(os.path.join(HERE, "iris_multinomial_path.npz"),
                        a0=fit["a0"], beta=fit["beta"], lambda_=fit["lambda"],
                        dev_ratio=fit["dev_ratio"], npasses=fit["npasses"], nulldev=fit["nulldev"],
                        args=np.array("family=multinomial alpha=0.8 nlambda=100 thresh=1e-3 "
                                      "maxit=1000 standardize=TRUE intercept=TRUE seed=1 mode=exact"))

    # ---- small sparse binomial epoch trajectories (exact and batched) ----
    from sgdnet_amd import data as D
    n, p, epochs = 2000, 150, 3
    pr = D.make_sparse_glm(n, p, 0.05, family="binomial", seed=21)
    X = D.as_scipy(pr)
    stream = po.Rng(21).stream(n, n * epochs)
    out = dict(ptr=pr["ptr"], idx=pr["idx"], val=pr["val"], y=pr["y"], stream=stream,
               gamma=0.03, alpha=2e-4, beta=3e-4)
    for tag, batch in (("exact", 0), ("batch64", 64)):
        st = po.new_state(1, p, n)
        po.saga(X, pr["y"], st, family="binomial", penalty="elasticnet", gamma=0.03, alpha=2e-4,
                beta=3e-4, max_iter=epochs, tol=0.0, stream=stream, batch=batch)
        out[f"w_{tag}"] = st["w"]
        out[f"b_{tag}"] = st["intercept"]
        out[f"G_{tag}"] = st["g_sum"]
    np.savez_compressed(os.path.join(HERE, "sparse_binomial_epochs.npz"), **out)
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
