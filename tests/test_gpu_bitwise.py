"""Exact mode, bit for bit.  The one-wavefront HIP exact kernels (sparse x; dense x with K*p <= 64) replay the
reference iteration in the reference's arithmetic order (-ffp-contract=off); with the plain-IEEE exp/log of
include/sgdnet_detmath.h in the family gradients on both sides, every state array after several epochs is
IDENTICAL to the CPU restatement built with -DORC_DET_MATH (oracle/liboracle_det.so) -- not close: equal.
The workgroup kernel for wider dense rows sums the dot product as a tree: equal to rounding."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
STATE = ("w", "intercept", "g_sum", "g_memory", "g_sum_intercept")


@pytest.fixture(scope="module")
def sa():
    import torch  # noqa: F401
    import sgdnet_amd
    if sgdnet_amd.load().sgdnet_device_count() < 1:
        pytest.fail("GPU tests need a HIP device; the backend has no CPU fallback")
    return sgdnet_amd


@pytest.fixture()
def det():
    from oracle import pyoracle as po
    po.use_det_math(True)
    yield po
    po.use_det_math(False)


def run(sa, po, x, y, *, family, K, penalty, gamma, alpha, beta, epochs, fit_intercept=True, c=None, tol=0.0):
    from test_gpu_parity import run_both
    return run_both(sa, po, x, y, family=family, K=K, penalty=penalty, gamma=gamma, alpha=alpha, beta=beta,
                    epochs=epochs, mode="exact", fit_intercept=fit_intercept, c=c, tol=tol, seed=3)


@pytest.mark.parametrize("family,K,penalty", [("binomial", 1, "elasticnet"), ("binomial", 1, "ridge"),
                                              ("multinomial", 3, "elasticnet"), ("multinomial", 10, "ridge"),
                                              ("gaussian", 1, "elasticnet"), ("mgaussian", 2, "grouplasso")])
@pytest.mark.parametrize("dense", [False, True])
def test_exact_kernels_are_bit_identical_to_the_det_oracle(sa, det, family, K, penalty, dense):
    from test_gpu_parity import make_problem
    if not dense:
        sa.set_option("exact_row_registers", 3)     # sparse x: the multi-wavefront kernels wherever they are legal
    if dense:
        x, y = make_problem(family, K, 900, 12 if K <= 3 else 6, None, seed=2, dense=True)   # K*p <= 64: the small kernel
    else:
        x, y = make_problem(family, K, 1500, 80, 0.06, seed=2)
    a, b = (1e-3, 0.0) if penalty == "ridge" else (5e-4, 5e-4)
    try:
        ref, got = run(sa, det, x, y, family=family, K=K, penalty=penalty, gamma=0.05, alpha=a, beta=b, epochs=4)
    finally:
        sa.set_option("exact_row_registers", 1)
    assert ref[0] == got[0]
    for name in STATE:
        assert np.array_equal(got[2][name], ref[2][name]), name


MULTI_CLASS_CASES = {
    # name: (family, K, penalty, n, p, density or row lengths, gamma, alpha, beta, epochs)
    "multinomial_many_features": ("multinomial", 3, "elasticnet", 6000, 4000, 0.002, 0.05, 1e-4, 2e-4, 3),
    "multinomial_ten_classes": ("multinomial", 10, "ridge", 3000, 2500, 0.004, 0.05, 1e-3, 0.0, 3),
    "mgaussian_group_lasso": ("mgaussian", 4, "grouplasso", 3000, 2000, 0.004, 0.03, 1e-4, 3e-4, 3),
    "multinomial_crowded": ("multinomial", 4, "elasticnet", 900, 20, 0.4, 0.05, 5e-4, 5e-4, 3),
    "multinomial_scale_reset": ("multinomial", 3, "elasticnet", 1200, 300, 0.03, 0.3, 1.0, 1e-3, 2),
}


@pytest.mark.parametrize("case", sorted(MULTI_CLASS_CASES))
@pytest.mark.parametrize("registers", [0, 1, 3])
def test_multi_wavefront_general_kernel_is_bit_identical(sa, det, case, registers):
    """Round 3: `saga_sparse_exact_mc_kernel` (the general iteration on eight wavefronts at once: features and sample of
    a draw registered in draw order, the intercepts handed on in draw order) against the det-math restatement -- equal,
    like the one-wavefront kernel (0); 1: where the host expects it to pay, 3: wherever legal (crowded rows, the
    mid-epoch rescaling)."""
    from test_gpu_parity import make_problem
    family, K, penalty, n, p, dens, gamma, alpha, beta, epochs = MULTI_CLASS_CASES[case]
    x, y = make_problem(family, K, n, p, dens, seed=13)
    with sa.option("exact_row_registers", registers):
        ref, got = run(sa, det, x, y, family=family, K=K, penalty=penalty, gamma=gamma, alpha=alpha, beta=beta, epochs=epochs)
    assert ref[0] == got[0]
    for name in STATE:
        assert np.array_equal(got[2][name], ref[2][name]), name


def test_wide_dense_kernel_and_convergence_epochs(sa, det):
    # K*p = 160: beyond the register-resident kernel.  The workgroup kernel (saga_dense_exact_wide_kernel)
    # adds the dot product up as a tree instead of feature by feature -- the one place where the order of
    # additions differs from the restatement -- so the states agree to rounding, not bit for bit
    from test_gpu_parity import make_problem, relerr
    x, y = make_problem("multinomial", 4, 700, 40, None, seed=5, dense=True)
    ref, got = run(sa, det, x, y, family="multinomial", K=4, penalty="elasticnet", gamma=0.02, alpha=1e-3, beta=1e-3,
                   epochs=300, tol=1e-3)
    assert ref[0] == got[0] and ref[0] < 300                                           # same stopping epoch
    for name in STATE:
        assert relerr(got[2][name], ref[2][name]) < 1e-10, name


def _ragged_problem(family, n, p, lengths, seed):
    """Sparse x (features x samples, as make_problem gives it) whose sample i holds lengths[i % len(lengths)]
    non-zeros: empty rows, rows of exactly one wavefront, rows past it."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    indptr, indices, data = [0], [], []
    for i in range(n):
        k = min(int(lengths[i % len(lengths)]), p)
        indices.append(np.sort(rng.choice(p, size=k, replace=False)))
        data.append(rng.standard_normal(k) / np.sqrt(max(k, 1)))
        indptr.append(indptr[-1] + k)
    x = sp.csc_matrix((np.concatenate(data), np.concatenate(indices).astype(np.int32), np.asarray(indptr, dtype=np.int64)),
                      shape=(p, n))
    lp = np.asarray(x.T @ rng.standard_normal(p)).ravel()
    if family == "binomial":
        y = (rng.random(n) < 1 / (1 + np.exp(-lp))).astype(float)
    else:
        y = lp + 0.1 * rng.standard_normal(n)
    return x, np.asfortranarray(y.reshape(1, n))


ROW_REGISTER_CASES = {
    # name: (family, penalty, n, p, density or row lengths, gamma, alpha, beta, epochs, fit_intercept)
    "crowded": ("binomial", "elasticnet", 800, 12, 0.5, 0.05, 5e-4, 5e-4, 4, True),       # neighbours share features
    "gaussian_ridge": ("gaussian", "ridge", 1500, 80, 0.06, 0.05, 1e-3, 0.0, 3, True),
    "group_of_one": ("gaussian", "grouplasso", 1200, 50, 0.1, 0.03, 1e-3, 2e-3, 3, False),
    "ragged": ("binomial", "elasticnet", 900, 200, (0, 3, 64, 65, 1, 130, 17, 64, 0, 200), 0.04, 5e-4, 1e-3, 3, True),
    "all_long": ("gaussian", "elasticnet", 400, 300, (120, 90, 70), 0.02, 1e-3, 1e-3, 2, True),
    "scale_reset": ("gaussian", "elasticnet", 1500, 60, 0.1, 0.3, 1.0, 1e-3, 2, True),    # wscale < SMALL every ~90 draws
    "scale_reset_ragged": ("gaussian", "elasticnet", 600, 150, (5, 80, 2, 64), 0.3, 1.0, 1e-3, 2, True),
    "state_in_memory": ("binomial", "elasticnet", 25000, 9000, 0.001, 0.05, 2e-5, 2e-5, 2, True),  # lags past the LDS cache
    # few non-zeros per row in many features (the several-consumer kernel by default), with a long row now and then
    "heavy_tail": ("binomial", "elasticnet", 4000, 6000, (5, 4, 6, 3, 5, 7, 4, 5, 120, 5, 6, 4, 5, 3, 70, 5, 4, 6, 5, 5), 0.04, 1e-4, 2e-4, 3, True),
}


@pytest.mark.parametrize("case", sorted(ROW_REGISTER_CASES))
@pytest.mark.parametrize("registers", [0, 1, 2, 3, 4])
def test_register_resident_sparse_kernel_is_bit_identical(sa, det, case, registers):
    """Round 3: `saga_sparse_exact_k1_kernel` (one response: the row's lanes keep w, g_sum and lag of their features
    in registers for the whole draw, the state of the next draw is requested a draw ahead and forwarded where two
    draws share a feature) and `saga_sparse_exact_k1m_kernel` (several consumer wavefronts, registration and the
    intercept chain in draw order) against the det-math restatement -- equal, like the general kernel (option
    exact_row_registers = 0) they replace; 1: the default choice, 2: one consumer with the state in memory, 3: several
    consumers wherever legal (also on the crowded and the long-row cases), 4: one consumer always."""
    from test_gpu_parity import make_problem
    family, penalty, n, p, shape, gamma, alpha, beta, epochs, fit_intercept = ROW_REGISTER_CASES[case]
    if isinstance(shape, tuple):
        x, y = _ragged_problem(family, n, p, shape, seed=11)
    else:
        x, y = make_problem(family, 1, n, p, shape, seed=7)
    with sa.option("exact_row_registers", registers):
        ref, got = run(sa, det, x, y, family=family, K=1, penalty=penalty, gamma=gamma, alpha=alpha, beta=beta,
                       epochs=epochs, fit_intercept=fit_intercept)
    assert ref[0] == got[0]
    for name in STATE:
        assert np.array_equal(got[2][name], ref[2][name]), name


@pytest.mark.parametrize("registers", [0, 1, 2, 3, 4])
def test_register_resident_kernel_with_repeated_draws_and_early_stop(sa, det, registers):
    # the same sample drawn several times in a row (its gradient memory and every feature are forwarded), and a
    # tolerance that stops the launch inside its block of epochs
    from test_gpu_parity import make_problem, run_both
    x, y = make_problem("binomial", 1, 700, 40, 0.15, seed=4)
    n = 700
    rng = np.random.default_rng(1)
    stream = np.repeat(rng.integers(0, n, size=n * 40 // 3 + 3), 3)[: n * 40].astype(np.uint32)
    with sa.option("exact_row_registers", registers):
        ref, got = run_both(sa, det, x, y, family="binomial", K=1, penalty="elasticnet", gamma=0.05, alpha=1e-3,
                            beta=1e-3, epochs=40, mode="exact", tol=1e-2, stream=stream)
    assert ref[0] == got[0] and ref[0] < 40
    for name in STATE:
        assert np.array_equal(got[2][name], ref[2][name]), name
