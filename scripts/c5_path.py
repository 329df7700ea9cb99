#!/usr/bin/env python3
"""BASELINE config 5 as stated: synthetic CSC 50M x 100 000, 0.01 % non-zeros, multinomial K = 10, a
warm-started 100-point lambda path (lambda.min.ratio 1e-4, alpha 0.5) on ONE MI355X, through sgdnet()
(mode = "auto": the binned kernels), reference semantics src/sgdnet.cpp:217-273.

    python scripts/c5_path.py [--n 50000000] [--nlambda 100] [--thresh 1e-3] [--out profiles/r04_c5_path.json]

Records wall time, epochs per lambda, return codes, monotonicity of dev.ratio and the multinomial
elastic-net KKT residual at the last lambda (tests/test_gpu_c5.py: _multinomial_kkt).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def multinomial_kkt(X, y, K, w, b, a_l2, b_l1, chunk=2_000_000):
    """Largest violation of the optimality conditions of (1/n) sum_i -log p_i,y_i + a/2 |w|^2 + b |w|_1
    (R/sgdnet.R:37-48, src/families.h:235-260), in sample chunks.  X: (p, n) CSC, sample i = column i."""
    p, n = X.shape
    g = np.zeros((K, p))
    icpt = np.zeros(K)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        Xc = X[:, lo:hi]
        lp = (Xc.T @ w.T) + b
        lp -= lp.max(axis=1, keepdims=True)
        P = np.exp(lp)
        P /= P.sum(axis=1, keepdims=True)
        P[np.arange(hi - lo), y[lo:hi].astype(np.int64)] -= 1.0
        g += (Xc @ P).T
        icpt += P.sum(axis=0)
    g = g / n + a_l2 * w
    viol = np.where(w == 0, np.maximum(np.abs(g) - b_l1, 0.0), np.abs(g + b_l1 * np.sign(w)))
    return float(viol.max()), float(np.abs(icpt).max() / n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50_000_000)
    ap.add_argument("--p", type=int, default=100_000)
    ap.add_argument("--nlambda", type=int, default=100)
    ap.add_argument("--first", type=int, default=0, help="run only the first FIRST lambdas of the path (0: all)")
    ap.add_argument("--thresh", type=float, default=1e-3)
    ap.add_argument("--maxit", type=int, default=1000)
    ap.add_argument("--kkt", action="store_true", help="KKT residual of the last lambda (a host pass over the matrix)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import sgdnet_amd as sa
    from sgdnet_amd import data as D
    K, dens, seed = 10, 1e-4, 5
    t0 = time.time()
    pr = D.make_sparse_glm(args.n, args.p, dens, family="multinomial", n_classes=K, seed=seed)
    X = D.as_scipy(pr)                      # (p, n): sample i = column i
    y = pr["y"].ravel()
    t_gen = time.time() - t0
    print(f"[c5] generated {args.n} x {args.p}, {X.nnz} non-zeros in {t_gen:.1f} s", file=sys.stderr, flush=True)
    xs = X.T.tocsc()                         # samples in rows, as sgdnet() takes it (R's dgCMatrix)
    lam = None
    if args.first:
        # the head of the 100-point path, written out: lambda_max = max_jk |x_j'(Y_k - mean Y_k)| / n / max(alpha, 0.001)
        # (src/families.h:300-325, src/utils.h:157-165), lambda_i log-spaced down to lambda_max * 1e-4 (src/math.h:42-56)
        cnt = np.bincount(y.astype(np.int64), minlength=K).astype(float)
        Yc = -np.tile(cnt / args.n, (args.n, 1))
        Yc[np.arange(args.n), y.astype(np.int64)] += 1.0
        lmax = float(np.abs(X @ Yc).max()) / args.n / 0.5
        del Yc
        grid = np.exp(np.log(lmax) + np.arange(args.nlambda) * (np.log(lmax * 1e-4) - np.log(lmax)) / (args.nlambda - 1))
        lam = grid[:args.first]
    os.environ.setdefault("SGDNET_TRACE", "1")
    t1 = time.time()
    fit = sa.sgdnet(xs, y, family="multinomial", alpha=0.5, nlambda=args.nlambda, lambda_min_ratio=1e-4, lambda_=lam,
                    standardize=False, thresh=args.thresh, maxit=args.maxit, mode="auto", seed=seed)
    wall = time.time() - t1
    dr = np.asarray(fit.dev_ratio)
    beta = np.stack([np.asarray(bk) for bk in fit.beta])      # list of K (p, n_lambda) arrays -> (K, p, n_lambda)
    rec = {
        "workload": f"C5: synthetic CSC {args.n}x{args.p}, {dens:.4%} nnz, multinomial K={K}, alpha=0.5, "
                    f"{args.nlambda}-point path, lambda.min.ratio=1e-4, warm starts, mode=auto, thresh={args.thresh}",
        "lambdas_run": int(len(fit.lambda_)),
        "wall_s": wall,
        "gen_s": t_gen,
        "npasses": float(fit.npasses),
        "return_codes": [int(c) for c in np.asarray(fit.return_codes).ravel()],
        "lambda": [float(v) for v in np.asarray(fit.lambda_).ravel()],
        "dev_ratio": [float(v) for v in dr.ravel()],
        "dev_ratio_monotone": bool(np.all(np.diff(dr) >= -1e-6)),
        "nonzero_last": int(np.count_nonzero(beta[:, :, -1])),
        "epochs_per_lambda_mean": float(fit.npasses) / max(1, len(fit.lambda_)),
    }
    if args.kkt:
        lam_last = float(fit.lambda_[-1])
        w = beta[:, :, -1]
        b = np.asarray(fit.a0)[:, -1]
        tk = time.time()
        kkt, icpt = multinomial_kkt(X, y, K, w, b, 0.5 * lam_last, 0.5 * lam_last)
        rec["kkt_last"] = {"lambda": lam_last, "residual": kkt, "residual_over_lambda": kkt / lam_last,
                           "intercept_residual": icpt, "seconds": time.time() - tk}
    line = json.dumps(rec)
    print(line, flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
