"""MatrixMarket writer / reader (host logic, CPU): round trips through scipy.io.mmread and back."""
import numpy as np
import scipy.io
import scipy.sparse as sp

from nalu_amd import mmio


def test_sparse_and_dense_round_trip(tmp_path):
    rng = np.random.RandomState(3)
    A = sp.random(7, 7, density=0.4, random_state=rng, format="csr")
    A.sort_indices()
    gids = np.array([3, 1, 7, 2, 5, 4, 6])  # local -> global (1-based), a permutation
    names = mmio.file_names(str(tmp_path / "continuity"), 2, 1, 0)
    assert names["mm"].endswith("continuity-O-2.mm.1") and names["rhs"].endswith("-O-2.rhs.1")
    mmio.write_sparse(names["mm"], 7, 7, gids, A.indptr, gids[A.indices], A.data, "ContinuityEQS",
                      "Tpetra matrix for: ContinuityEQS")
    B = scipy.io.mmread(names["mm"]).toarray()
    P = np.zeros((7, 7))
    P[gids - 1, np.arange(7)] = 1.0  # global <- local
    assert np.array_equal(B, P @ A.toarray() @ P.T)  # 17 digits: exact round trip
    b = rng.uniform(-1, 1, 7)
    mmio.write_dense(names["rhs"], b)
    assert np.array_equal(scipy.io.mmread(names["rhs"]).ravel(), b)
    # multi-rank names carry the rank
    assert mmio.file_names("x", 0, 2, 1)["mm"] == "x-O-0.mm.2.1"
    # the reader gives back exactly what was written (entry order kept)
    nr, nc, r, c, v = mmio.read_sparse(names["mm"])
    rows = np.repeat(np.arange(7), np.diff(A.indptr))
    assert (nr, nc) == (7, 7) and np.array_equal(r, gids[rows]) and np.array_equal(c, gids[A.indices])
    assert np.array_equal(v, A.data)
    assert np.array_equal(mmio.read_dense(names["rhs"]), b)
    import pytest
    with pytest.raises(ValueError):
        mmio.read_sparse(names["rhs"])
    with pytest.raises(ValueError):
        mmio.read_dense(names["mm"])
