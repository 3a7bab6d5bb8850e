"""CPU: the sampled-row checker of the full-size GPU tests (tests/rowcheck.py) is itself checked
against the oracle on a small problem -- every user and item row, with and without the bias
slots, ratings and -binary-data.  What it recomputes is step A of hgaprec.cc:1340-1366."""
import numpy as np
import pytest
import torch

from tests import rowcheck
from tests.util import make_problem


@pytest.mark.parametrize("hier,bias,binary", [(True, False, False), (True, True, False), (False, False, True), (False, True, False)])
def test_rowcheck_recomputation_equals_the_oracle(orc, hier, bias, binary):
    n, m, K = 120, 90, 7
    rowptr, col, val = make_problem(n, m, 2500, seed=3, heavy_user=True, heavy_item=True, singles=True)
    M = orc.Model(n, m, K, hier, bias, binary)
    M.set_csr(rowptr, col, np.ones_like(val) if binary else val)
    M.initialize(5)
    el_t, el_b = torch.from_numpy(M.state("THETA_ELOG")), torch.from_numpy(M.state("BETA_ELOG"))
    ub = torch.from_numpy(M.state("UBIAS_ELOG")) if bias else None
    ib = torch.from_numpy(M.state("IBIAS_ELOG")) if bias else None
    rp, c = torch.from_numpy(rowptr), torch.from_numpy(col.astype(np.int32))
    v = None if binary else torch.from_numpy(val)
    wu = rowcheck.expected_user_rows(list(range(n)), rp, c, v, el_t, el_b, ub, ib).numpy()
    wi = rowcheck.expected_item_rows(list(range(m)), rp, c, v, el_t, el_b, ub, ib).numpy()
    M.iterate(1)
    gu, gi = M.state("THETA_SHAPE") - 0.3, M.state("BETA_SHAPE") - 0.3
    if bias:
        gu = np.concatenate([gu, (M.state("UBIAS_SHAPE") - 0.3)[:, None]], 1)
        gi = np.concatenate([gi, (M.state("IBIAS_SHAPE") - 0.3)[:, None]], 1)
    assert np.max(np.abs(wu - gu) / np.maximum(np.abs(gu), 1e-12)) < 1e-11
    assert np.max(np.abs(wi - gi) / np.maximum(np.abs(gi), 1e-12)) < 1e-11


def test_pick_rows_covers_the_named_cases():
    deg_u = torch.tensor([3, 0, 600, 5, 1200, 7, 8, 2000, 1, 2] * 30)
    deg_i = torch.tensor([10, 5000, 40, 41, 39, 3, 2, 1, 0, 9000] * 20)
    wi = {"heavy_min_nnz_item": 40, "tile_rows_item": 64, "tile_rows_user": 50, "heavy_min_nnz_user": 0}
    users, items = rowcheck.pick_rows(deg_u, deg_i, wi, n_users=20, n_items=10, seed=1)
    assert 0 in users and 299 in users and 63 in users and 64 in users           # first, last, a tile boundary of the item pass
    assert any(deg_u[u] > 512 for u in users) and any(deg_u[u] == 0 for u in users)
    assert 199 in items and any(deg_i[i] == 9000 for i in items)
    assert any(deg_i[i] in (39, 40, 41) for i in items) and 49 in items and 50 in items
