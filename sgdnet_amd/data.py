"""Synthetic sparse GLM problems of the BASELINE.json shapes.

Per sample: z_i = max(1, Binomial(p, density)) distinct feature ids (ascending), values
N(0,1); planted beta with 10 % non-zeros; binomial y ~ Bernoulli(sigmoid(x.beta)),
multinomial y ~ Categorical(softmax(B x)).  The generator is blocked so that a rank can
produce only its own shard of the global problem (same bytes whatever the world size).
"""
import numpy as np

BLOCK = 1 << 20  # samples per generator block (seeded independently)


def planted_beta(p, K, seed):
    rng = np.random.Generator(np.random.PCG64([seed, 0xBE7A]))
    beta = np.where(rng.random((K, p)) < 0.1, rng.standard_normal((K, p)), 0.0)
    return beta


def _gen_block(lo, hi, p, density, family, beta, seed):
    rng = np.random.Generator(np.random.PCG64([seed, lo // BLOCK + 1]))
    m = hi - lo
    z = np.maximum(1, rng.binomial(p, density, size=m)).astype(np.int64)
    ptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(z, out=ptr[1:])
    nnz = int(ptr[-1])
    rows = np.repeat(np.arange(m, dtype=np.int64), z)
    cols = rng.integers(0, p, size=nnz, dtype=np.int64)
    # sort (row, col); drop the rare duplicate feature id inside a row
    key = rows * p + cols
    key.sort()
    keep = np.ones(nnz, dtype=bool)
    keep[1:] = key[1:] != key[:-1]
    key = key[keep]
    rows = key // p
    cols = (key - rows * p).astype(np.int32)
    vals = rng.standard_normal(key.size)
    counts = np.bincount(rows, minlength=m)
    ptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(counts, out=ptr[1:])
    K = beta.shape[0]
    lp = np.empty((K, m))
    for k in range(K):
        lp[k] = np.bincount(rows, weights=vals * beta[k, cols], minlength=m)
    if family == "binomial":
        y = (rng.random(m) < 1.0 / (1.0 + np.exp(-lp[0]))).astype(np.float64)
    elif family == "multinomial":
        g = rng.gumbel(size=(K, m))
        y = np.argmax(lp + g, axis=0).astype(np.float64)
    elif family == "gaussian":
        y = lp[0] + 0.1 * rng.standard_normal(m)
    else:
        y = (lp + 0.1 * rng.standard_normal((K, m)))
    return ptr, cols, vals, y


def make_sparse_glm(n, p, density, family="binomial", n_classes=1, seed=0, lo=0, hi=None):
    """Samples [lo, hi) of the n x p problem, sample-major.

    Returns dict(ptr int64[m+1], idx int32[nnz], val float64[nnz], y (Ky, m) F-order,
    n_total=n).  scipy view: csc_matrix((val, idx, ptr), shape=(p, m)).
    """
    hi = n if hi is None else hi
    K = n_classes
    beta = planted_beta(p, K, seed)
    ptrs, idxs, vals, ys = [np.zeros(1, dtype=np.int64)], [], [], []
    base = 0
    start = (lo // BLOCK) * BLOCK
    for b0 in range(start, hi, BLOCK):
        b1 = min(b0 + BLOCK, n)
        ptr, cols, v, y = _gen_block(b0, b1, p, density, family, beta, seed)
        s0, s1 = max(lo, b0) - b0, min(hi, b1) - b0
        q0, q1 = ptr[s0], ptr[s1]
        ptrs.append(ptr[s0 + 1:s1 + 1] - q0 + base)
        idxs.append(cols[q0:q1])
        vals.append(v[q0:q1])
        ys.append(y[..., s0:s1])
        base += q1 - q0
    ptr = np.concatenate(ptrs)
    y = np.concatenate(ys, axis=-1)
    y = np.asfortranarray(y.reshape(-1, hi - lo))
    return dict(ptr=ptr, idx=np.concatenate(idxs), val=np.concatenate(vals), y=y, n_total=n,
                p=p, beta_true=beta)


def as_scipy(prob):
    import scipy.sparse as sp
    m = prob["ptr"].size - 1
    return sp.csc_matrix((prob["val"], prob["idx"], prob["ptr"]), shape=(prob["p"], m))


def algorithmic_bytes(row_nnz, stream, K, Ky=1):
    """S_i = 16 + 12 z_i + 8 Ky + 16 K bytes per inner iteration
    (two row pointers, z indices, z values, y, gradient memory read + write),
    summed over the draws of `stream` (= 16 + 12 z + 16 K for Ky == K == 1 ... )."""
    z = row_nnz[stream].astype(np.float64)
    return float(np.sum(8.0 + 12.0 * z + 8.0 * Ky + 16.0 * K))


def step_size(max_sq_norm, alpha_l2, fit_intercept, family, n):
    """StepSize of the reference (src/utils.h:31-51)."""
    L_scaling = 1.0 if family in ("gaussian", "mgaussian") else 0.25
    L = (max_sq_norm + float(fit_intercept)) * L_scaling + alpha_l2
    return 1.0 / (2.0 * L + min(L, 2.0 * n * alpha_l2))
