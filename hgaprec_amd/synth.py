"""Synthetic sparse rating matrices -- generator G(seed, n, m, nnz, alpha_u, alpha_i)
of SURVEY.md section 8(d), and the bench-mode initial state.

Runs on whatever torch device it is given (the GPU in bench.py, the CPU in the
tests); torch is plumbing here (sort / unique / searchsorted on 5e7 keys), the
product path starts at hpf_upload_csr.
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


CHUNK_DRAWS = 1_500_000_000


def _degrees(n, m, nnz, alpha_u, gen, device):
    """power-law user degrees d_u ~ (rank+1)^-alpha_u, 1 <= d_u <= m/2,
    sum(d) = nnz (when the caps allow), ranks randomly permuted"""
    cap = max(1, m // 2)
    w = torch.arange(1, n + 1, dtype=torch.float64, device=device) ** (-alpha_u)
    t = w / w.sum() * nnz
    d = torch.clamp(torch.floor(t), 1, cap).to(torch.int64)
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
                frac = (t - torch.floor(t))[idx]
                d[idx[torch.argsort(frac, descending=True)[:rem]]] += 1
        else:
            idx = torch.nonzero(d > 1, as_tuple=False).flatten()
            if idx.numel() == 0:
                break
            if -rem >= idx.numel():
                d[idx] -= torch.clamp(d[idx] - 1, max=(-rem) // idx.numel())
            else:
                d[idx[torch.argsort(d[idx], descending=True)[:(-rem)]]] -= 1
    perm = torch.randperm(n, generator=gen, device=device)
    return d[perm]


def generate(n, m, nnz, alpha_u=0.5, alpha_i=0.8, seed=0, device="cpu", binary=False,
             item_seed=None, topup_rounds=4):
    """-> rowptr int64[n+1], col uint32[nnz'], val uint8[nnz'] (numpy, host).
    Users get power-law degrees, items are drawn without replacement per user
    from a power-law popularity (draw, dedupe, top up); columns sorted in a row.
    item_seed fixes the item popularity permutation independently of `seed`
    (multi-GPU: every rank shares the items, owns its users)."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    igen = torch.Generator(device=device)
    igen.manual_seed(int(seed if item_seed is None else item_seed) + 7919)
    d = _degrees(n, m, nnz, alpha_u, gen, device)
    p = torch.arange(1, m + 1, dtype=torch.float64, device=device) ** (-alpha_i)
    cdf = torch.cumsum(p / p.sum(), 0)
    iperm = torch.randperm(m, generator=igen, device=device)

    # users are processed in contiguous chunks of at most CHUNK_DRAWS draws (the
    # device sort behind torch.unique takes < 2^31 keys); one chunk -- every
    # configuration up to C3's 1e9 nonzeros -- is the unchunked algorithm
    cum = torch.cumsum(d, 0)
    bounds = [0]
    while bounds[-1] < n:
        base = int(cum[bounds[-1] - 1]) if bounds[-1] > 0 else 0
        nxt = int(torch.searchsorted(cum, torch.tensor(base + CHUNK_DRAWS, device=device), right=True))
        bounds.append(min(n, max(nxt, bounds[-1] + 1)))
    counts = torch.zeros(n, dtype=torch.int64, device=device)
    cols, vals = [], []
    pr = torch.tensor(RATING_P, dtype=torch.float64, device=device)
    for a, b in zip(bounds[:-1], bounds[1:]):
        users = torch.arange(a, b, device=device, dtype=torch.int64)
        keys = torch.empty(0, dtype=torch.int64, device=device)
        dc = d[a:b]
        need = dc.clone()
        for _ in range(topup_rounds):
            tot = int(need.sum())
            if tot == 0:
                break
            u = torch.repeat_interleave(users, need)
            r = torch.rand(tot, generator=gen, device=device, dtype=torch.float64)
            it = iperm[torch.searchsorted(cdf, r).clamp(max=m - 1)]
            keys = torch.unique(torch.cat([keys, u * m + it]))
            del u, r, it
            have = torch.bincount(keys // m - a, minlength=b - a)
            need = (dc - have).clamp(min=0)
        counts[a:b] = torch.bincount(keys // m - a, minlength=b - a)
        cols.append((keys % m).to(torch.int32).cpu().numpy().view(np.uint32))
        if not binary:
            vals.append((torch.multinomial(pr, keys.numel(), replacement=True, generator=gen) + 1)
                        .to(torch.uint8).cpu().numpy())
        del keys
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(counts, 0)
    rp = rowptr.cpu().numpy()
    c = cols[0] if len(cols) == 1 else np.concatenate(cols)
    v = None if binary else (vals[0] if len(vals) == 1 else np.concatenate(vals))
    return rp, c, v


def heldout(n, m, cnt, seed, device="cpu", binary=False):
    """cnt held-out pairs, sorted by (user, item) like std::map<Rating,int>"""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed) + 104729)
    keys = torch.unique(torch.randint(0, n * m, (cnt,), generator=gen, device=device, dtype=torch.int64))
    pr = torch.tensor(RATING_P, dtype=torch.float64, device=device)
    y = (torch.multinomial(pr, keys.numel(), replacement=True, generator=gen) + 1).to(torch.int32)
    if binary:
        y = torch.ones_like(y)
    return ((keys // m).cpu().numpy().astype(np.uint32), (keys % m).cpu().numpy().astype(np.uint32),
            y.cpu().numpy())


def initial_state(rows, K, seed, device="cpu", prior_v=None):
    """bench-mode start (no parity claim -- the parity path draws MT19937 on the
    host): shape = 0.3 + 0.01 U, E = shape / (0.3 + 0.1 U'), Elog = psi(shape) -
    log(rate).  prior_v: returns the xi/eta style vector start instead
    (shape 0.3 + 0.01 U, rate 0.3 + prior_v)."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    if prior_v is not None:
        s = 0.3 + 0.01 * torch.rand(rows, generator=gen, device=device, dtype=torch.float64)
        r = torch.full_like(s, 0.3 + float(prior_v))
    else:
        s = 0.3 + 0.01 * torch.rand(rows, K, generator=gen, device=device, dtype=torch.float64)
        r = 0.3 + 0.1 * torch.rand(rows, K, generator=gen, device=device, dtype=torch.float64)
    e = s / r
    el = torch.special.digamma(s) - torch.log(r)
    return dict(shape=s.cpu().numpy(), rate=r.cpu().numpy(), E=e.cpu().numpy(), Elog=el.cpu().numpy())
