"""Synthetic workload generator and the algorithmic byte model."""
import numpy as np

from sgdnet_amd import data as D


def test_shards_are_slices_of_the_global_problem():
    n, p = 3000, 50
    full = D.make_sparse_glm(n, p, 0.1, family="binomial", seed=5)
    lo, hi = 1000, 2200
    part = D.make_sparse_glm(n, p, 0.1, family="binomial", seed=5, lo=lo, hi=hi)
    q0, q1 = full["ptr"][lo], full["ptr"][hi]
    assert np.array_equal(part["ptr"], full["ptr"][lo:hi + 1] - q0)
    assert np.array_equal(part["idx"], full["idx"][q0:q1])
    assert np.array_equal(part["val"], full["val"][q0:q1])
    assert np.array_equal(part["y"], full["y"][:, lo:hi])


def test_rows_are_sorted_distinct_and_nonempty():
    pr = D.make_sparse_glm(5000, 40, 0.05, family="multinomial", n_classes=4, seed=2)
    ptr, idx = pr["ptr"], pr["idx"]
    assert np.all(np.diff(ptr) >= 1)
    for i in range(0, 5000, 97):
        row = idx[ptr[i]:ptr[i + 1]]
        assert np.all(np.diff(row) > 0) and row.min() >= 0 and row.max() < 40
    assert set(np.unique(pr["y"])) <= {0.0, 1.0, 2.0, 3.0}


def test_algorithmic_bytes_model():
    # S_i = 16 + 12 z_i + 16 K for K = Ky = 1 -> 152 B at z = 10
    row_nnz = np.array([10, 3, 20])
    stream = np.array([0, 0, 2, 1], dtype=np.uint32)
    want = sum(16 + 12 * z + 16 for z in (10, 10, 20, 3))
    assert D.algorithmic_bytes(row_nnz, stream, 1) == want
    assert D.algorithmic_bytes(np.array([10]), np.array([0]), 1) == 152.0
    # K = 10 multinomial: 16 + 12 z + 16 K = 296
    assert D.algorithmic_bytes(np.array([10]), np.array([0]), 10) == 296.0
