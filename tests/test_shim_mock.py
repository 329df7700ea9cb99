"""shim/sgdnet_shim.c compiled against the R-API mock (tests/rmock): what can be checked without
a GPU -- the shared object loads, R_init_sgdnet registers exactly the reference's two .Call
routines with arity 3 (src/RcppExports.cpp:36-45), the control list is read by name, and failures
come back as R errors (here: no HIP device, since the backend has no CPU fallback)."""
import os

import numpy as np
import pytest

import rshim


@pytest.fixture()
def R():
    if not os.path.exists(rshim.SO):
        pytest.fail(f"{rshim.SO} missing: run ./build.sh")
    L = rshim.lib()
    L.rmock_reset()
    L.R_init_sgdnet(None)
    return L


def test_registration_matches_the_reference(R):
    import ctypes as C
    got = []
    for i in range(3):
        name, nargs = C.create_string_buffer(64), C.c_int(0)
        if R.rmock_registered(i, name, 64, C.byref(nargs)):
            got.append((name.value.decode(), nargs.value))
    assert got == [("_sgdnet_SgdnetDense", 3), ("_sgdnet_SgdnetSparse", 3)]
    assert R.rmock_dynamic_symbols() == 0                    # R_useDynamicSymbols(dll, FALSE)
    for sym in ("_sgdnet_SgdnetDense", "_sgdnet_SgdnetSparse", "R_init_sgdnet"):
        assert hasattr(R, sym)


def test_mock_rng_is_rs_generator(R):
    R.rmock_set_seed(1)                                      # set.seed(1); runif(3)
    assert np.allclose([R.unif_rand() for _ in range(3)], [0.2655087, 0.3721239, 0.5728534], atol=5e-8)
    R.rmock_set_seed(42)
    assert np.allclose([R.unif_rand() for _ in range(2)], [0.9148060, 0.9370754], atol=5e-8)


def test_control_fields_are_looked_up_by_name(R):
    x = rshim.r_matrix(np.zeros((4, 2)))
    y = rshim.r_matrix(np.zeros((4, 1)))
    ctl = rshim.control_list(family="gaussian", drop="tol")
    with pytest.raises(rshim.RError, match="control list lacks field 'tol'"):
        rshim.call("_sgdnet_SgdnetDense", x, y, ctl)
    assert R.rmock_protect_depth() == 0
    with pytest.raises(rshim.RError, match="unknown family 'poisson'"):
        rshim.call("_sgdnet_SgdnetDense", x, y, rshim.control_list(family="poisson"))


def test_argument_shape_errors_are_r_errors(R):
    ctl = rshim.control_list(family="gaussian")
    with pytest.raises(rshim.RError, match="must match"):
        rshim.call("_sgdnet_SgdnetDense", rshim.r_matrix(np.zeros((4, 2))), rshim.r_matrix(np.zeros((3, 1))), ctl)
    with pytest.raises(rshim.RError, match="x must be a matrix"):
        rshim.call("_sgdnet_SgdnetDense", rshim.r_real(np.zeros(8)), rshim.r_matrix(np.zeros((4, 1))), ctl)
    with pytest.raises(rshim.RError, match='no slot of name "Dim"'):
        rshim.call("_sgdnet_SgdnetSparse", rshim.lib().rmock_s4(), rshim.r_matrix(np.zeros((4, 1))), ctl)


def test_backend_failure_becomes_an_r_error_after_the_rng_scope_closed(R):
    import sgdnet_amd
    if sgdnet_amd.load().sgdnet_device_count() > 0:
        pytest.skip("a HIP device is present: the end-to-end tests in test_gpu_shim.py apply")
    rng = np.random.default_rng(0)
    x = rshim.r_matrix(rng.standard_normal((20, 3)))
    y = rshim.r_matrix(rng.standard_normal((20, 1)))
    R.rmock_set_seed(1)
    with pytest.raises(rshim.RError, match="no HIP device"):
        rshim.call("_sgdnet_SgdnetDense", x, y, rshim.control_list(family="gaussian"))
    # GetRNGstate(); [PutRNGstate(): the generator state is handed to the backend]; ...; PutRNGstate()
    assert R.rmock_rng_scope_calls() == 102
    assert R.rmock_protect_depth() == 0
