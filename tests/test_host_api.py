"""Host-side mirror of sgdnet.default(): argument validation happens before the native
call (reference R/sgdnet.R:211-339; tests/testthat/test-assertions.R), so it is testable
without a GPU."""
import numpy as np
import pytest

import sgdnet_amd as sa


@pytest.fixture
def xy():
    rng = np.random.default_rng(0)
    return rng.normal(size=(30, 3)), rng.normal(size=30)


def test_argument_errors(xy):
    x, y = xy
    with pytest.raises(ValueError, match="number of samples"):
        sa.sgdnet(x, y[:-1])
    with pytest.raises(ValueError, match="mixing parameter"):
        sa.sgdnet(x, y, alpha=1.5)
    with pytest.raises(ValueError, match="must be positive"):
        sa.sgdnet(x, y, lambda_=[-1.0])
    with pytest.raises(ValueError, match="zero length"):
        sa.sgdnet(x, y, lambda_=[])
    with pytest.raises(ValueError, match="cannot be negative"):
        sa.sgdnet(x, y, thresh=-1)
    with pytest.raises(ValueError, match="negative or zero"):
        sa.sgdnet(x, y, maxit=0)
    with pytest.raises(ValueError, match="NA values"):
        sa.sgdnet(x, np.where(np.arange(30) == 3, np.nan, y))
    with pytest.raises(ValueError, match="must be logical"):
        sa.sgdnet(x, y, intercept=1)
    with pytest.raises(ValueError, match="should be one of"):
        sa.sgdnet(x, y, family="poisson")


def test_family_specific_errors(xy):
    x, y = xy
    with pytest.raises(ValueError, match="one-dimensional"):
        sa.sgdnet(x, np.c_[y, y], family="gaussian")
    with pytest.raises(ValueError, match="more than two classes"):
        sa.sgdnet(x, np.arange(30) % 3, family="binomial")
    with pytest.raises(ValueError, match="only one class"):
        sa.sgdnet(x, np.zeros(30), family="binomial")
    with pytest.raises(ValueError, match="only has 1 observations"):
        sa.sgdnet(x, np.r_[1, np.zeros(29)], family="binomial")
    with pytest.raises(ValueError, match="only two classes"):
        sa.sgdnet(x, np.arange(30) % 2, family="multinomial")
    with pytest.raises(ValueError, match="must not be one-dimensional"):
        sa.sgdnet(x, y, family="mgaussian")


def test_shard_bounds_partition():
    from sgdnet_amd.parallel import shard_bounds
    for n, w in ((10, 3), (10_000_000, 8), (7, 7), (5, 1)):
        edges = [shard_bounds(n, w, r) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        sizes = [hi - lo for lo, hi in edges]
        assert max(sizes) - min(sizes) <= 1
