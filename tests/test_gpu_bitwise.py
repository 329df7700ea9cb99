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
    if dense:
        x, y = make_problem(family, K, 900, 12 if K <= 3 else 6, None, seed=2, dense=True)   # K*p <= 64: the small kernel
    else:
        x, y = make_problem(family, K, 1500, 80, 0.06, seed=2)
    a, b = (1e-3, 0.0) if penalty == "ridge" else (5e-4, 5e-4)
    ref, got = run(sa, det, x, y, family=family, K=K, penalty=penalty, gamma=0.05, alpha=a, beta=b, epochs=4)
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
