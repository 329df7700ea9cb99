"""SURVEY.md 8 row f3/b end to end: `.Call(_sgdnet_SgdnetDense|Sparse, x, y, control)` through the real
shim code (shim/sgdnet_shim.c, compiled against the R-API mock of tests/rmock because the image has
no R) into libsgdnet_hip.so on the GPU, compared with the CPU oracle's fit of the same problem under
the same set.seed(): the returned list's names, order, shapes and unlist(beta) layout
(src/sgdnet.cpp:275-284, consumed by R/sgdnet.R:368-431), and the random draws taken from R's
generator (Rcpp::RNGScope + one R::runif per inner iteration, src/RcppExports.cpp:14,27)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import rshim

pytestmark = pytest.mark.gpu
NAMES = ["a0", "beta", "losses", "npasses", "nulldev", "dev.ratio", "lambda", "return_codes"]


@pytest.fixture()
def R():
    import torch  # noqa: F401
    import sgdnet_amd
    if sgdnet_amd.load().sgdnet_device_count() < 1:
        pytest.fail("GPU tests need a HIP device; the backend has no CPU fallback")
    L = rshim.lib()
    L.rmock_reset()
    L.R_init_sgdnet(None)
    return L


def problem(family, n, p, K, seed, sparse):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, p))
    if sparse:
        x *= rng.random((n, p)) < 0.3
    B = rng.standard_normal((p, max(K, 1)))
    lp = x @ B
    if family == "gaussian":
        y = lp[:, 0] + 0.3 * rng.standard_normal(n)
    elif family == "binomial":
        y = (rng.random(n) < 1 / (1 + np.exp(-lp[:, 0]))).astype(float)
    elif family == "multinomial":
        y = np.argmax(lp + rng.gumbel(size=lp.shape), axis=1).astype(float)
    else:
        y = lp + 0.3 * rng.standard_normal(lp.shape)
    return (sp.csc_matrix(x) if sparse else x), y


def mock_state(R):
    st = (C.c_uint32 * 625)()
    R.rmock_rng_state(st)
    return np.array(st[:])


@pytest.mark.parametrize("rng_path", ["handoff", "callback"])
@pytest.mark.parametrize("family,K,sparse,alpha", [("gaussian", 1, False, 1.0), ("binomial", 1, True, 0.5),
                                                   ("multinomial", 3, False, 0.8), ("mgaussian", 2, True, 0.5),
                                                   ("binomial", 1, False, 0.0)])
def test_call_through_the_shim_matches_the_oracle(R, family, K, sparse, alpha, rng_path):
    """rng_path: "handoff" = the default with R's Mersenne-Twister -- .Random.seed is handed to the backend,
    which continues R's stream on the device and hands the advanced state back; "callback" =
    options(sgdnet.rng = "callback"): one unif_rand() per inner iteration, as Rcpp's R::runif."""
    from oracle import pyoracle as po
    n, p, nl = 300, 6, 12
    x, y = problem(family, n, p, K, 5, sparse)
    ctl = rshim.control_list(family=family, alpha=alpha, n_classes=K, nlambda=nl, lambda_min_ratio=1e-3,
                             thresh=1e-4, is_sparse=sparse)
    ry = rshim.r_matrix(np.asarray(y, dtype=float).reshape(n, -1))
    if rng_path == "callback":
        rshim.set_option("sgdnet.rng", "callback")
    R.rmock_set_seed(7)
    res = rshim.call("_sgdnet_SgdnetSparse" if sparse else "_sgdnet_SgdnetDense",
                     rshim.r_dgcmatrix(x) if sparse else rshim.r_matrix(x), ry, ctl)
    got = rshim.decode_result(res)
    orng = po.Rng(7)
    ref = po.fit(x, y, family=family, alpha=alpha, nlambda=nl, lambda_min_ratio=1e-3, thresh=1e-4, n_classes=K,
                 rng=orng)
    # structure R/sgdnet.R:368-431 relies on
    assert got["names"] == NAMES
    assert got["beta_dims"] == [(K, p)] * nl and got["a0"].shape == (K, nl) and got["losses"] == []
    # values: below lambda_max (at lambda_max itself the stopping epoch hangs on libm's last bit)
    assert np.allclose(got["lambda_"], ref["lambda"], rtol=1e-12)
    assert abs(got["nulldev"] - ref["nulldev"]) <= 1e-10 * abs(ref["nulldev"])
    if got["npasses"] == ref["npasses"]:
        # unlist(beta): lambda-major, then feature, class fastest == the oracle's (K, p, L) Fortran order
        scale = np.abs(ref["beta"]).max()
        assert np.abs(got["unlist_beta"] - ref["beta"].ravel(order="F")).max() <= 1e-9 * scale
        assert np.abs(got["a0"] - ref["a0"]).max() <= 1e-9 * max(1.0, np.abs(ref["a0"]).max())
        assert np.allclose(got["dev_ratio"], ref["dev_ratio"], atol=1e-9)
        assert np.array_equal(got["return_codes"], ref["return_codes"])
        # R's generator was advanced exactly as the reference advances it
        assert np.array_equal(mock_state(R)[1:], np.array(orng.state.mt[:]))
    else:
        assert family in ("binomial", "multinomial")       # only exp/log can move a stopping epoch
        assert np.allclose(got["dev_ratio"], ref["dev_ratio"], atol=5e-3)
    if rng_path == "callback":
        assert R.rmock_unif_count() == int(got["npasses"]) * n   # one unif_rand() per inner iteration
        assert R.rmock_rng_scope_calls() == 101                  # GetRNGstate(); ...; PutRNGstate();
    else:
        assert R.rmock_unif_count() == 0                         # no host call per draw ...
        host = po.Rng(7)                                         # ... and R's generator where the reference leaves it
        host.stream(n, int(got["npasses"]) * n)
        R.GetRNGstate()
        assert [R.unif_rand() for _ in range(5)] == list(host.unif(5))
    assert R.rmock_protect_depth() == 0


def test_integer_inputs_are_coerced_like_rcpp_as(R):
    from oracle import pyoracle as po
    rng = np.random.default_rng(3)
    n, p = 200, 4
    x = rng.integers(-3, 4, size=(n, p))
    y = rng.integers(0, 20, size=n)
    ctl = rshim.control_list(family="gaussian", nlambda=8, thresh=1e-5)
    R.rmock_set_seed(2)
    res = rshim.call("_sgdnet_SgdnetDense", rshim.r_matrix(x, integer=True), rshim.r_int(y), ctl)   # y without dim
    got = rshim.decode_result(res)
    ref = po.fit(x.astype(float), y.astype(float), family="gaussian", nlambda=8, thresh=1e-5, rng=po.Rng(2))
    assert got["npasses"] == ref["npasses"]
    assert np.abs(got["unlist_beta"] - ref["beta"].ravel(order="F")).max() <= 1e-9 * np.abs(ref["beta"]).max()


def test_debug_losses_grow_with_the_epochs_not_with_maxit(R):
    from oracle import pyoracle as po
    n, p, nl = 250, 5, 6
    x, y = problem("binomial", n, p, 1, 9, False)
    # maxit = 1e8 is the reference's benchmark setting (data-raw/benchmarks.R): n_lambda x max_iter doubles
    # would be 4.8 GB here (80 GB for a 100-lambda path)
    ctl = rshim.control_list(family="binomial", alpha=0.5, nlambda=nl, lambda_min_ratio=1e-2, thresh=1e-3,
                             maxit=1e8, debug=True)
    R.rmock_set_seed(4)
    got = rshim.decode_result(rshim.call("_sgdnet_SgdnetDense", rshim.r_matrix(x), rshim.r_matrix(y.reshape(n, 1)), ctl))
    ref = po.fit(x, y, family="binomial", alpha=0.5, nlambda=nl, lambda_min_ratio=1e-2, thresh=1e-3, maxit=2000,
                 debug=True, rng=po.Rng(4))
    assert len(got["losses"]) == nl and sum(len(l) for l in got["losses"]) == int(got["npasses"])
    for i in range(1, nl):
        if len(got["losses"][i]) == len(ref["losses"][i]):
            assert np.allclose(got["losses"][i], ref["losses"][i], rtol=1e-9)
    assert all(np.isfinite(l).all() and (l > 0).all() for l in got["losses"])      # tests/testthat/test-options.R


def test_mode_option_reaches_the_batched_kernels_with_virtual_shards(R):
    """options(sgdnet.mode = "auto"): the shim's unif_rand() draws are laid out per virtual shard by
    the driver; the fit must reach the optimum the exact iteration reaches."""
    from sgdnet_amd import data as D
    n, p = 200_000, 100
    pr = D.make_sparse_glm(n, p, 0.05, family="binomial", seed=21)
    x = D.as_scipy(pr).T.tocsc()
    y = pr["y"].reshape(n, 1)
    kw = dict(family="binomial", alpha=0.5, lambda_=[1e-3], standardize=False, thresh=1e-9, maxit=300, is_sparse=True)
    rshim.set_option("sgdnet.mode", "auto")
    R.rmock_set_seed(1)
    fast = rshim.decode_result(rshim.call("_sgdnet_SgdnetSparse", rshim.r_dgcmatrix(x), rshim.r_matrix(y),
                                          rshim.control_list(**kw)))
    assert fast["return_codes"][0] == 0 and R.rmock_unif_count() == 0
    import sgdnet_amd as sa
    ref = sa.sgdnet(x, y.ravel(), family="binomial", alpha=0.5, lambda_=[1e-3], standardize=False, thresh=1e-9,
                    maxit=300, mode="batched", seed=5)
    assert np.abs(fast["unlist_beta"] - ref.beta[:, 0]).max() <= 1e-6 * np.abs(ref.beta).max()


def test_other_rng_kinds_keep_the_callback(R):
    """RNGkind("Wichmann-Hill") etc.: the backend cannot continue that stream, the shim keeps unif_rand()."""
    n, p = 200, 3
    x, y = problem("gaussian", n, p, 1, 2, False)
    R.rmock_set_seed(1)
    R.rmock_set_rng_kind(10400)                               # kind code % 100 != 3: not Mersenne-Twister
    R.PutRNGstate()
    got = rshim.decode_result(rshim.call("_sgdnet_SgdnetDense", rshim.r_matrix(x), rshim.r_matrix(y.reshape(n, 1)),
                                         rshim.control_list(family="gaussian", nlambda=5)))
    assert R.rmock_unif_count() == int(got["npasses"]) * n
