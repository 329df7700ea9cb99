"""Jump-ahead of R's Mersenne-Twister (sgdnet_amd/csrc/mt_jump.cpp), host side: no GPU needed."""
import ctypes as C

import numpy as np
import pytest

import sgdnet_amd as sa


@pytest.mark.parametrize("J", [1, 2, 623, 624, 625, 19937, 123_457, 10_000_000])
def test_jump_polynomial_moves_the_state_down_the_same_stream(J):
    L = sa.load()
    poly = (C.c_uint32 * 624)()
    assert L.sgdnet_rng_jump_poly(C.c_uint64(J), poly) == 0
    a, b = sa.RRng(1), sa.RRng(1)
    a.unif(700), b.unif(700)                    # a state in the middle of a regenerated block
    out = sa.RRng(0)
    L.sgdnet_rng_jump(C.byref(a.state), poly, C.byref(out.state))
    b.stream(10, J)                             # J draws, one by one
    assert np.array_equal(out.unif(2000), b.unif(2000))


def test_jump_from_a_fresh_seed_and_composition():
    L = sa.load()
    p1, p2 = (C.c_uint32 * 624)(), (C.c_uint32 * 624)()
    assert L.sgdnet_rng_jump_poly(C.c_uint64(5000), p1) == 0 and L.sgdnet_rng_jump_poly(C.c_uint64(12345), p2) == 0
    s0 = sa.RRng(42)                            # straight after set.seed(): mti = 624, nothing regenerated yet
    a, b = sa.RRng(0), sa.RRng(0)
    L.sgdnet_rng_jump(C.byref(s0.state), p1, C.byref(a.state))
    L.sgdnet_rng_jump(C.byref(a.state), p2, C.byref(b.state))
    ref = sa.RRng(42)
    ref.stream(3, 5000 + 12345)
    assert np.array_equal(b.unif(1000), ref.unif(1000))
