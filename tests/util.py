"""Shared helpers for the test suite (problem builders, oracle <-> device glue)."""
from __future__ import annotations

import numpy as np


def make_problem(n, m, nnz_target, seed, alpha_u=0.6, alpha_i=0.8, max_rating=5,
                 heavy_user=False, heavy_item=False, singles=False):
    """Small power-law CSR in the reference's visiting order (rows = users in
    seq order, entries in 'file order' = random order inside a row)."""
    rng = np.random.default_rng(seed)
    du = (np.arange(n) + 1.0) ** -alpha_u
    du = np.maximum(1, np.round(du / du.sum() * nnz_target)).astype(np.int64)
    du = np.minimum(du, max(1, m // 2))
    rng.shuffle(du)
    if heavy_user:
        du[rng.integers(n)] = m            # one user rated everything
    if singles:
        du[: max(1, n // 10)] = 1
    pi = (np.arange(m) + 1.0) ** -alpha_i
    pi /= pi.sum()
    pi = pi[rng.permutation(m)]
    rows = []
    for u in range(n):
        d = int(du[u])
        if d >= m:
            items = rng.permutation(m)
        else:
            items = rng.choice(m, size=d, replace=False, p=pi)
        rows.append(items.astype(np.uint32))
    if heavy_item:                          # one item rated by every user
        hi = np.uint32(rng.integers(m))
        rows = [r if hi in r else np.concatenate([r, [hi]]).astype(np.uint32) for r in rows]
    # make sure every item appears at least once (the reference registers items
    # only when seen, so its m equals the number of distinct items)
    seen = np.zeros(m, bool)
    for r in rows:
        seen[r] = True
    missing = np.flatnonzero(~seen)
    for k, it in enumerate(missing):
        u = k % n
        rows[u] = np.concatenate([rows[u], [it]]).astype(np.uint32)
    rowptr = np.zeros(n + 1, np.int64)
    rowptr[1:] = np.cumsum([len(r) for r in rows])
    col = np.concatenate(rows).astype(np.uint32)
    p = np.array([0.06, 0.11, 0.26, 0.35, 0.22])[:max_rating]
    val = rng.choice(np.arange(1, max_rating + 1), size=col.size, p=p / p.sum()).astype(np.uint8)
    return rowptr, col, val


def heldout_pairs(n, m, cnt, seed, max_rating=5):
    rng = np.random.default_rng(seed)
    key = np.unique(rng.integers(0, n * m, size=cnt, dtype=np.int64))
    u = (key // m).astype(np.uint32)
    i = (key % m).astype(np.uint32)
    y = rng.integers(1, max_rating + 1, size=key.size).astype(np.int32)
    return u, i, y           # sorted by (u, i) like std::map<Rating,int>


INIT_STATES_HIER = ["THETA_SHAPE", "THETA_E", "THETA_ELOG", "BETA_SHAPE", "BETA_E", "BETA_ELOG",
                    "XI_SHAPE", "XI_RATE", "XI_E", "XI_ELOG", "ETA_SHAPE", "ETA_RATE", "ETA_E", "ETA_ELOG"]
INIT_STATES_FLAT = ["THETA_SHAPE", "THETA_E", "THETA_ELOG", "BETA_SHAPE", "BETA_E", "BETA_ELOG"]
INIT_STATES_BIAS = ["UBIAS_SHAPE", "UBIAS_E", "UBIAS_ELOG", "IBIAS_SHAPE", "IBIAS_E", "IBIAS_ELOG"]


def init_states(hier, bias):
    s = list(INIT_STATES_HIER if hier else INIT_STATES_FLAT)
    if bias:
        s += INIT_STATES_BIAS
    return s


def compare_states(hier, bias):
    s = ["THETA_SHAPE", "THETA_RATE", "THETA_E", "THETA_ELOG",
         "BETA_SHAPE", "BETA_RATE", "BETA_E", "BETA_ELOG"]
    if hier:
        s += ["XI_SHAPE", "XI_RATE", "XI_E", "XI_ELOG", "ETA_SHAPE", "ETA_RATE", "ETA_E", "ETA_ELOG"]
    if bias:
        s += ["UBIAS_SHAPE", "UBIAS_RATE", "UBIAS_E", "UBIAS_ELOG",
              "IBIAS_SHAPE", "IBIAS_RATE", "IBIAS_E", "IBIAS_ELOG"]
    return s


def copy_state(oracle_model, dev, hier, bias):
    """start the device model from the oracle's (reference-order MT19937) state"""
    for w in init_states(hier, bias):
        dev.set_state(w, oracle_model.state(w))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0
