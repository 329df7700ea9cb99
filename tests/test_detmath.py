"""include/sgdnet_detmath.h: the plain-IEEE exp/log shared by the HIP exact kernels and the det build of the
oracle.  Host side: accuracy against libm and the special values; device == host is what
tests/test_gpu_bitwise.py establishes through the kernels."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po


@pytest.fixture(scope="module")
def L():
    po.use_det_math(True)
    lib = po.lib()
    po.use_det_math(False)
    lib.orc_det_exp.restype = lib.orc_det_log.restype = C.c_double
    lib.orc_det_exp.argtypes = lib.orc_det_log.argtypes = [C.c_double]
    return lib


def ulps(a, b):
    return np.abs(a - b) / np.spacing(np.abs(b))


def test_exp_accuracy_and_specials(L):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-40, 40, 200_000), rng.uniform(-745, 709, 50_000), rng.uniform(-1e-3, 1e-3, 50_000)])
    got = np.array([L.orc_det_exp(v) for v in x])
    assert ulps(got, np.exp(x)).max() <= 1.0
    assert L.orc_det_exp(0.0) == 1.0 and L.orc_det_exp(710.0) == np.inf and L.orc_det_exp(-800.0) == 0.0
    assert np.isnan(L.orc_det_exp(np.nan)) and L.orc_det_exp(np.inf) == np.inf and L.orc_det_exp(-np.inf) == 0.0
    assert L.orc_det_exp(1e-20) == 1.0 and L.orc_det_exp(-740.0) == pytest.approx(np.exp(-740.0), rel=1e-3)   # subnormal result


def test_log_accuracy_and_specials(L):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(1, 64, 200_000), 1 + rng.uniform(0, 1e-3, 50_000), rng.uniform(1e-300, 1e300, 50_000),
                        rng.uniform(0.5, 1.0, 50_000), np.array([5e-324, 1e-310, 2.0, 0.5])])
    got = np.array([L.orc_det_log(v) for v in x])
    assert ulps(got, np.log(x)).max() <= 2.0
    assert L.orc_det_log(1.0) == 0.0 and L.orc_det_log(0.0) == -np.inf and np.isnan(L.orc_det_log(-1.0))
    assert L.orc_det_log(np.inf) == np.inf and np.isnan(L.orc_det_log(np.nan))
