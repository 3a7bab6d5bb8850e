"""Sampled-row value check of a phi pass at ANY size (test infrastructure; also the
`self_check` of bench.py).

Mass conservation cannot see a wrong-row gather: a nonzero's phi sums to max(y, 1)
whichever row of the other side was read.  So for a sample of OWNER rows the raw phi
sums are recomputed here in plain fp64 (torch on whatever device holds the data; no
library code, no oracle) straight from the definition the reference uses,

    x_k   = Elog_theta[u, k] + Elog_beta[i, k]   (+ the two bias slots)   hgaprec.cc:206-239
    phi_k = max(y, 1) * softmax(x)_k                                      hgaprec.cc:1355-1356
    S_user[u, :] = sum_i phi,   S_item[i, :] = sum_u phi                  hgaprec.cc:1357-1366

and compared with what the device wrote (shape - 0.3).  An item with 10^5 raters is a
10^5 x K gather: milliseconds on the GPU, seconds in numpy.
"""
from __future__ import annotations

import numpy as np
import torch

PIECE = 1 << 20            # nonzeros of one owner row handled at a time (bounds the [d x K] temporaries)


def _as_i64(t):
    return t.to(torch.int64) if t.dtype != torch.int64 else t


def _phi_sums(x_own, elog_oth, idx, yy, own_bias, oth_bias):
    """sum over the nonzeros `idx` (rows of the other side) of yy * softmax([x_own + elog_oth[idx], biases]).
    -> (K sums, own-bias-slot sum or None)"""
    K = x_own.numel()
    acc = torch.zeros(K, dtype=torch.float64, device=x_own.device)
    accb = torch.zeros((), dtype=torch.float64, device=x_own.device)
    for a in range(0, idx.numel(), PIECE):
        j = idx[a:a + PIECE]
        x = elog_oth[j] + x_own[None, :]                       # [d, K]
        if own_bias is not None:
            xb = torch.stack([own_bias.expand(j.numel()), oth_bias[j]], 1)       # the owner's slot first
            x = torch.cat([x, xb], 1)
        p = torch.softmax(x, 1) * yy[a:a + PIECE, None]
        acc += p[:, :K].sum(0)
        if own_bias is not None:
            accb += p[:, K].sum()
    return acc, (accb if own_bias is not None else None)


def item_nonzeros(rowptr, col, items, chunk=1 << 28):
    """positions (ascending) of the nonzeros of each item in `items`, found by scanning the CSR's
    column array in slices (torch.nonzero does not take 2^31 elements at once) -> {item: int64 positions}"""
    dev = col.device
    sel = torch.as_tensor(sorted(set(int(i) for i in items)), dtype=col.dtype, device=dev)
    pos = []
    for a in range(0, col.numel(), chunk):
        c = col[a:a + chunk]
        hit = torch.isin(c, sel).nonzero().flatten()
        if hit.numel():
            pos.append(hit + a)
    pos = torch.cat(pos) if pos else torch.empty(0, dtype=torch.int64, device=dev)
    its = col[pos]
    return {int(i): pos[its == i] for i in sel.tolist()}


def expected_user_rows(users, rowptr, col, val, elog_t, elog_b, ubias=None, ibias=None):
    """raw phi sums of the user rows `users` -> float64 [len(users), K (+1: the user-bias slot)]"""
    out = []
    for u in users:
        a, b = int(rowptr[u]), int(rowptr[u + 1])
        idx = _as_i64(col[a:b])
        yy = torch.ones(b - a, dtype=torch.float64, device=col.device) if val is None else \
            torch.clamp(val[a:b], min=1).to(torch.float64)
        s, sb = _phi_sums(elog_t[u], elog_b, idx, yy, None if ubias is None else ubias[u], ibias)
        out.append(s if sb is None else torch.cat([s, sb[None]]))
    return torch.stack(out) if out else torch.empty(0, elog_t.shape[1], dtype=torch.float64)


def expected_item_rows(items, rowptr, col, val, elog_t, elog_b, ubias=None, ibias=None):
    """raw phi sums of the item rows `items` -> float64 [len(items), K (+1: the item-bias slot)]"""
    where = item_nonzeros(rowptr, col, items)
    out = []
    for i in items:
        pos = where[int(i)]
        users = torch.searchsorted(rowptr, pos, right=True) - 1
        yy = torch.ones(pos.numel(), dtype=torch.float64, device=col.device) if val is None else \
            torch.clamp(val[pos], min=1).to(torch.float64)
        s, sb = _phi_sums(elog_b[i], elog_t, users, yy, None if ibias is None else ibias[i], ubias)
        out.append(s if sb is None else torch.cat([s, sb[None]]))
    return torch.stack(out) if out else torch.empty(0, elog_b.shape[1], dtype=torch.float64)


def pick_rows(deg_u, deg_i, wi=None, n_users=500, n_items=50, seed=0, seg_max=512):
    """owner rows worth checking: random ones, the last, rows cut into several 512-nonzero
    segments, the heaviest (two-level combine), rows either side of the heavy / light bar of a
    tiled side, rows at the tile boundaries of the gathered matrix.  deg_*: int64 degrees (torch)."""
    rng = np.random.default_rng(seed)
    wi = wi or {}

    def side(deg, cnt, bar, tile_rows_as_gathered):
        n = int(deg.numel())
        pick = set(int(x) for x in rng.integers(0, n, size=min(cnt, n)))
        pick |= {0, n - 1}
        d = deg.cpu().numpy()
        order = np.argsort(d, kind="stable")
        pick |= set(int(x) for x in order[-3:])                        # heaviest rows
        longr = np.flatnonzero(d > seg_max)
        if longr.size:                                                 # straddle a 512-segment cut
            by = longr[np.argsort(d[longr], kind="stable")]
            pick |= set(int(x) for x in by[:3]) | set(int(x) for x in by[-2:])
        if bar:                                                        # either side of the heavy / light bar
            k = int(np.searchsorted(d[order], bar))
            pick |= set(int(x) for x in order[max(0, k - 3):k + 3])
        if tile_rows_as_gathered:                                      # first / last row of a tile
            T = int(tile_rows_as_gathered)
            for t in (1, 2, max(1, (n // T) // 2), max(1, n // T - 1)):
                for r in (t * T - 1, t * T):
                    if 0 <= r < n:
                        pick.add(int(r))
        empty = np.flatnonzero(d == 0)
        if empty.size:
            pick.add(int(empty[0]))
        return sorted(pick)

    users = side(deg_u, n_users, wi.get("heavy_min_nnz_user", 0), wi.get("tile_rows_item", 0))
    items = side(deg_i, n_items, wi.get("heavy_min_nnz_item", 0), wi.get("tile_rows_user", 0))
    return users, items


def item_degrees(col, m, chunk=1 << 29):
    d = torch.zeros(m, dtype=torch.int64, device=col.device)
    for a in range(0, col.numel(), chunk):
        d += torch.bincount(col[a:a + chunk].to(torch.int64), minlength=m)
    return d


def check_handle(D, rowptr, col, val, bias=False, n_users=500, n_items=50, seed=0, rtol=1e-9, users=None, items=None):
    """Runs ONE iteration on the handle from whatever state it holds and checks the phi sums of a
    sample of user and item rows against the fp64 recomputation from the exported Elog arrays.
    rowptr / col / val: torch tensors of the CSR the handle was fed (same device as the recomputation).
    -> dict(max_rel_err, rows_checked, ok, worst_row)"""
    dev = col.device

    def get(w):
        return D.get_state_device(w) if dev.type == "cuda" else torch.from_numpy(D.get_state(w))

    el_t, el_b = get("THETA_ELOG"), get("BETA_ELOG")
    ub = ib = None
    if bias:
        ub, ib = get("UBIAS_ELOG"), get("IBIAS_ELOG")
    m = el_b.shape[0]
    if users is None or items is None:
        deg_u = rowptr[1:] - rowptr[:-1]
        deg_i = item_degrees(col, m)
        pu, pi = pick_rows(deg_u, deg_i, D.work_info(), n_users, n_items, seed)
        users = pu if users is None else users
        items = pi if items is None else items
    want_u = expected_user_rows(users, rowptr, col, val, el_t, el_b, ub, ib)
    want_i = expected_item_rows(items, rowptr, col, val, el_t, el_b, ub, ib)
    del el_t, el_b
    if D_single(D):
        D.iterate(1)
    else:
        D.iterate_local()
        D.iterate_global()
    iu = torch.as_tensor(users, dtype=torch.int64, device=dev)
    ii = torch.as_tensor(items, dtype=torch.int64, device=dev)
    got_u = get("THETA_SHAPE")[iu] - 0.3
    got_i = get("BETA_SHAPE")[ii] - 0.3
    if bias:
        got_u = torch.cat([got_u, (get("UBIAS_SHAPE")[iu] - 0.3)[:, None]], 1)
        got_i = torch.cat([got_i, (get("IBIAS_SHAPE")[ii] - 0.3)[:, None]], 1)

    def err(got, want):
        if not want.numel():
            return 0.0, -1
        e = ((got - want).abs() / torch.clamp(want.abs(), min=1e-12)).amax(1)
        k = int(torch.argmax(e))
        return float(e[k]), k

    eu, ku = err(got_u, want_u)
    ei, ki = err(got_i, want_i)
    worst = ("user", users[ku]) if eu >= ei else ("item", items[ki])
    return {"max_rel_err": max(eu, ei), "max_rel_err_users": eu, "max_rel_err_items": ei,
            "rows_checked": {"users": len(users), "items": len(items)}, "worst_row": list(worst),
            "ok": bool(max(eu, ei) < rtol)}


def D_single(D):
    """the handle iterates on its own (one rank) -- a shard handle of several ranks is stepped by
    iterate_local + iterate_global with its own sums only (what a 1-of-N shard run on its own does)"""
    return getattr(D, "n_ranks", 1) == 1
