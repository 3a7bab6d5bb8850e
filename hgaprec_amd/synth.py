"""Synthetic sparse rating matrices -- generator G(seed, n, m, nnz, alpha_u, alpha_i)
of SURVEY.md section 8(d), and the bench-mode initial state.

Every random quantity is a pure function of (seed, user, draw index) or of
(seed, user, item) through a counter hash (splitmix64), never of a stateful
generator: any contiguous user range of the SAME matrix can therefore be
produced on its own, on any device, by any rank -- which is what the strong-
scaling bench needs (each rank builds only its shard of BASELINE config C3).

Runs on whatever torch device it is given (the GPU in bench.py, the CPU in the
tests); torch is plumbing here (sort / unique / searchsorted on 1e9 keys), the
product path starts at hpf_upload_csr / hpf_upload_csr_device.
"""
from __future__ import annotations

import numpy as np
import torch

RATING_P = (0.06, 0.11, 0.26, 0.35, 0.22)      # MovieLens-like, ratings 1..5

# (n, m, nnz, K, alpha_u, alpha_i, seed, flags) of BASELINE.md section 4
CONFIGS = {
    "C1": dict(n=6040, m=3681, nnz=800_167, K=20, alpha_u=0.6, alpha_i=0.8, seed=20260901,
               hier=True, bias=False, binary=False),
    "C2": dict(n=1_000_000, m=100_000, nnz=50_000_000, K=100, alpha_u=0.5, alpha_i=0.8,
               seed=20260902, hier=True, bias=False, binary=False),
    "C3": dict(n=10_000_000, m=1_000_000, nnz=1_000_000_000, K=100, alpha_u=0.5, alpha_i=0.8,
               seed=20260903, hier=True, bias=False, binary=False),
    "C4": dict(n=480_189, m=17_770, nnz=100_000_000, K=200, alpha_u=0.7, alpha_i=1.0,
               seed=20260904, hier=True, bias=True, binary=False),
    "C5": dict(n=50_000_000, m=2_000_000, nnz=5_000_000_000, K=50, alpha_u=0.9, alpha_i=1.1,
               seed=20260905, hier=True, bias=False, binary=True),
}

CHUNK_DRAWS = 1_500_000_000        # the device sort behind torch.unique takes < 2^31 keys

_M64 = (1 << 64) - 1


def _s64(x: int) -> int:
    """python int -> the int64 with the same low 64 bits"""
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


_C1, _C2, _G = _s64(0xBF58476D1CE4E5B9), _s64(0x94D049BB133111EB), _s64(0x9E3779B97F4A7C15)


def _lsr(x, k):
    """logical shift right of an int64 tensor"""
    return (x >> k) & ((1 << (64 - k)) - 1)


def _mix(x):
    """splitmix64 finaliser on int64 tensors (wrapping arithmetic)"""
    x = (x ^ _lsr(x, 30)) * _C1
    x = (x ^ _lsr(x, 27)) * _C2
    return x ^ _lsr(x, 31)


def _hash(seed: int, a, b=None):
    """64 well-mixed bits per element of the int64 tensor(s) a (, b)"""
    x = _mix(a * _G + _s64(seed * 0x632BE59BD9B4E019 + 0x2545F4914F6CDD1D))
    if b is not None:
        x = _mix(x ^ (b * _C2 + _G))
    return x


def _u01(h):
    """top 53 bits of a hash as a double in [0, 1)"""
    return _lsr(h, 11).to(torch.float64) * (1.0 / 9007199254740992.0)


def _perm(n, seed, device):
    """a permutation of 0..n-1 that depends only on (n, seed)"""
    h = _hash(seed, torch.arange(n, dtype=torch.int64, device=device))
    return torch.sort(h, stable=True)[1]


def _powerlaw_cpu(count, alpha):
    """(rank+1)^-alpha for rank < count and its sum, in float64 ON THE HOST (numpy): the
    floating-point part of the generator must not depend on which device runs the rest
    -- pow and the order of a parallel sum differ between back ends"""
    w = np.arange(1, count + 1, dtype=np.float64) ** (-float(alpha))
    return w, float(np.sum(w))


def degrees(n, m, nnz, alpha_u, seed, device="cpu"):
    """power-law user degrees d_u ~ (rank+1)^-alpha_u, 1 <= d_u <= m/2,
    sum(d) = nnz (when the caps allow), ranks permuted by a hash of the seed;
    int64[n].  The real-valued targets are computed on the host, the rest is
    integer arithmetic: identical on every device / rank."""
    device = torch.device(device)
    cap = max(1, m // 2)
    w, wsum = _powerlaw_cpu(n, alpha_u)
    t_h = w / wsum * float(nnz)
    del w
    fl_h = np.floor(t_h)
    frac_all = torch.from_numpy(t_h - fl_h).to(device)
    d = torch.clamp(torch.from_numpy(fl_h).to(device), 1, cap).to(torch.int64)
    del t_h, fl_h
    for _ in range(16):
        rem = int(nnz) - int(d.sum())
        if rem == 0:
            break
        if rem > 0:
            idx = torch.nonzero(d < cap, as_tuple=False).flatten()
            if idx.numel() == 0:
                break
            if rem >= idx.numel():
                d[idx] += torch.clamp(cap - d[idx], max=rem // idx.numel())
            else:
                frac = frac_all[idx]
                d[idx[torch.sort(frac, descending=True, stable=True)[1][:rem]]] += 1
        else:
            idx = torch.nonzero(d > 1, as_tuple=False).flatten()
            if idx.numel() == 0:
                break
            if -rem >= idx.numel():
                d[idx] -= torch.clamp(d[idx] - 1, max=(-rem) // idx.numel())
            else:
                d[idx[torch.sort(d[idx], descending=True, stable=True)[1][:(-rem)]]] -= 1
    return d[_perm(n, seed, device)]


def item_cdf(m, alpha_i, device="cpu"):
    """cumulative popularity of the items by rank, float64[m], built on the host in a
    fixed (sequential) order and copied to `device`"""
    p, psum = _powerlaw_cpu(m, alpha_i)
    return torch.from_numpy(np.cumsum(p / psum)).to(torch.device(device))


def generate_device(n, m, nnz, alpha_u=0.5, alpha_i=0.8, seed=0, device="cpu", binary=False,
                    item_seed=None, topup_rounds=4, user_range=None, deg=None, fill=True):
    """-> torch tensors on `device`: rowptr int64[b-a+1], col int32[nnz'],
    val uint8[nnz'] | None for users [a, b) = user_range (default: all) of the
    matrix G(seed, n, m, nnz, alpha_u, alpha_i).  Users get power-law degrees,
    items are drawn without replacement per user from a power-law popularity
    (draw, dedupe, top up `topup_rounds` times; what a user still lacks after that
    -- heavy users, whose draws keep hitting the same popular items -- is filled
    with the most popular items the user does not have yet, so that every user
    reaches the planned degree and the matrix the planned nnz); columns sorted
    inside a row; col holds item ids < m < 2^31.  Every user's row is a function
    of (seed, user) alone: user ranges generate independently.  `deg` may pass
    in degrees(...) already computed."""
    device = torch.device(device)
    a, b = (0, n) if user_range is None else (int(user_range[0]), int(user_range[1]))
    d_all = degrees(n, m, nnz, alpha_u, seed, device) if deg is None else deg.to(device)
    d = d_all[a:b]
    del d_all
    cdf = item_cdf(m, alpha_i, device)
    iseed = int(seed if item_seed is None else item_seed) + 7919
    iperm = _perm(m, iseed, device)
    nloc = b - a

    # users are processed in contiguous chunks of at most CHUNK_DRAWS draws
    cum = torch.cumsum(d, 0)
    bounds = [0]
    while bounds[-1] < nloc:
        base = int(cum[bounds[-1] - 1]) if bounds[-1] > 0 else 0
        nxt = int(torch.searchsorted(cum, torch.tensor(base + CHUNK_DRAWS, device=device), right=True))
        bounds.append(min(nloc, max(nxt, bounds[-1] + 1)))
    del cum
    counts = torch.zeros(nloc, dtype=torch.int64, device=device)
    cols, vals = [], []
    thr = torch.cumsum(torch.tensor(RATING_P, dtype=torch.float64, device=device), 0)
    for ca, cb in zip(bounds[:-1], bounds[1:]):
        users = torch.arange(a + ca, a + cb, device=device, dtype=torch.int64)     # global user ids
        keys = torch.empty(0, dtype=torch.int64, device=device)
        dc = d[ca:cb]
        need = dc.clone()
        drawn = torch.zeros_like(dc)                       # draws made so far per user (the counter)
        for _ in range(topup_rounds):
            tot = int(need.sum())
            if tot == 0:
                break
            u = torch.repeat_interleave(users, need)
            first = torch.cumsum(need, 0) - need           # offset of each user's run
            q = torch.arange(tot, device=device, dtype=torch.int64) - torch.repeat_interleave(first - drawn, need)
            r = _u01(_hash(seed, u, q))
            del q
            it = iperm[torch.searchsorted(cdf, r).clamp(max=m - 1)]
            del r
            keys = torch.unique(torch.cat([keys, u * m + it]))
            del u, it
            drawn = drawn + need
            have = torch.bincount(keys // m - (a + ca), minlength=cb - ca)
            need = (dc - have).clamp(min=0)
        if fill and int(need.sum()) > 0:
            keys = _fill_by_popularity(keys, users, dc, need, iperm, m, a + ca)
        counts[ca:cb] = torch.bincount(keys // m - (a + ca), minlength=cb - ca)
        cols.append((keys % m).to(torch.int32))
        if not binary:
            rr = _u01(_hash(seed + 104723, keys))          # rating = f(seed, user, item)
            vals.append((torch.searchsorted(thr, rr).clamp(max=len(RATING_P) - 1) + 1).to(torch.uint8))
            del rr
        del keys
    rowptr = torch.zeros(nloc + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(counts, 0)
    col = cols[0] if len(cols) == 1 else torch.cat(cols)
    val = None if binary else (vals[0] if len(vals) == 1 else torch.cat(vals))
    return rowptr, col, val


def _fill_by_popularity(keys, users, dc, need, iperm, m, user0, piece=400_000_000):
    """users that still lack `need` items after the rejection rounds get the most popular
    items they do not have yet.  The first d_u items in popularity order hold at most
    (d_u - need_u) of the user's present items, hence at least need_u new ones: candidates
    are those d_u items, the new ones among them are ranked, the first need_u are kept.
    keys: sorted unique user * m + item of the chunk; returns the same with the fill."""
    device = keys.device
    sel = torch.nonzero(need > 0, as_tuple=False).flatten()
    out = [keys]
    # needy users in pieces of bounded candidate count
    L = dc[sel]
    cum = torch.cumsum(L, 0)
    start = 0
    while start < sel.numel():
        base = int(cum[start - 1]) if start > 0 else 0
        stop = int(torch.searchsorted(cum, torch.tensor(base + piece, device=device), right=True))
        stop = min(sel.numel(), max(stop, start + 1))
        ss = sel[start:stop]
        Ls = dc[ss]
        tot = int(Ls.sum())
        first = torch.cumsum(Ls, 0) - Ls
        owner = torch.repeat_interleave(torch.arange(ss.numel(), device=device), Ls)
        rank = torch.arange(tot, device=device, dtype=torch.int64) - first[owner]
        cand = users[ss][owner] * m + iperm[rank]
        del rank
        if keys.numel():
            pos = torch.searchsorted(keys, cand).clamp(max=keys.numel() - 1)
            new = keys[pos] != cand
            del pos
        else:
            new = torch.ones_like(cand, dtype=torch.bool)
        c = torch.cumsum(new.to(torch.int64), 0)
        before = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), c])[first]   # new ones before the user's run
        order = c - before[owner]                                 # 1-based index among the user's new candidates
        keep = new & (order <= need[ss][owner])
        out.append(cand[keep])
        del cand, new, c, order, keep, owner
        start = stop
    return torch.unique(torch.cat(out))


def generate(n, m, nnz, alpha_u=0.5, alpha_i=0.8, seed=0, device="cpu", binary=False,
             item_seed=None, topup_rounds=4, user_range=None):
    """generate_device(...) copied to the host: rowptr int64, col uint32, val uint8 | None (numpy)"""
    rp, c, v = generate_device(n, m, nnz, alpha_u, alpha_i, seed, device, binary, item_seed,
                               topup_rounds, user_range)
    return rp.cpu().numpy(), c.cpu().numpy().view(np.uint32), None if v is None else v.cpu().numpy()


def heldout(n, m, cnt, seed, device="cpu", binary=False):
    """cnt held-out pairs, sorted by (user, item) like std::map<Rating,int>"""
    device = torch.device(device)
    idx = torch.arange(cnt, dtype=torch.int64, device=device)
    keys = torch.unique(_lsr(_hash(seed + 104729, idx), 1) % (n * m))
    thr = torch.cumsum(torch.tensor(RATING_P, dtype=torch.float64, device=device), 0)
    y = (torch.searchsorted(thr, _u01(_hash(seed + 15485863, keys))).clamp(max=len(RATING_P) - 1) + 1).to(torch.int32)
    if binary:
        y = torch.ones_like(y)
    return ((keys // m).cpu().numpy().astype(np.uint32), (keys % m).cpu().numpy().astype(np.uint32),
            y.cpu().numpy())


def initial_state_device(rows, K, seed, device="cpu", prior_v=None, row0=0):
    """bench-mode start (no parity claim -- the parity path draws MT19937 on the
    host): shape = 0.3 + 0.01 U, E = shape / (0.3 + 0.1 U'), Elog = psi(shape) -
    log(rate), U and U' hashes of (seed, global row, column).  prior_v: the
    xi/eta style vector start instead (shape 0.3 + 0.01 U, rate 0.3 + prior_v).
    Rows [row0, row0 + rows) of the full array: shards of one state agree.
    -> dict of float64 torch tensors on `device`."""
    device = torch.device(device)
    r = torch.arange(row0, row0 + rows, dtype=torch.int64, device=device)
    if prior_v is not None:
        s = 0.3 + 0.01 * _u01(_hash(seed, r))
        rt = torch.full_like(s, 0.3 + float(prior_v))
    else:
        e = r[:, None] * K + torch.arange(K, dtype=torch.int64, device=device)[None, :]
        s = 0.3 + 0.01 * _u01(_hash(seed, e))
        rt = 0.3 + 0.1 * _u01(_hash(seed + 1_000_003, e))
        del e
    return dict(shape=s, rate=rt, E=s / rt, Elog=torch.special.digamma(s) - torch.log(rt))


def initial_state(rows, K, seed, device="cpu", prior_v=None, row0=0):
    """initial_state_device(...) as numpy arrays on the host"""
    st = initial_state_device(rows, K, seed, device, prior_v, row0)
    return {k: v.cpu().numpy() for k, v in st.items()}
