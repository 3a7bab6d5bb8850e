"""The synthetic ratings generator (hgaprec_amd/synth.py): the matrix BASELINE's
configs are drawn from.  CPU-only: its floating-point parts run on the host by
construction, so what is checked here holds on the GPU box as well."""
import numpy as np
import torch

from hgaprec_amd import synth


def test_heavy_tailed_config_reaches_the_planned_nnz():
    # C5-like exponents on a small grid: the heaviest users' degrees (up to m / 2) cannot
    # be reached by rejection sampling in a few rounds -- the popularity fill must close it
    n, m, nnz = 4000, 300, 200_000
    rp, c, v = synth.generate_device(n, m, nnz, 0.9, 1.1, seed=5)
    assert int(rp[-1]) == nnz
    d = synth.degrees(n, m, nnz, 0.9, 5)
    assert torch.equal(rp[1:] - rp[:-1], d) and int(d.max()) == m // 2
    rpn, cn = rp.numpy(), c.numpy()
    for u in range(0, n, 37):
        r = cn[rpn[u]:rpn[u + 1]]
        assert (np.diff(r) > 0).all() and r.min() >= 0 and r.max() < m      # sorted, unique, in range
    assert v.min() >= 1 and v.max() <= 5
    # rejection alone falls short here (what VERDICT r2 #4 found at C4 / C5)
    rp0, _, _ = synth.generate_device(n, m, nnz, 0.9, 1.1, seed=5, topup_rounds=4, fill=False)
    assert int(rp0[-1]) < nnz


def test_user_ranges_generate_independently():
    n, m, nnz = 3000, 500, 120_000
    rp, c, v = synth.generate_device(n, m, nnz, 0.8, 1.0, seed=11)
    for a, b in ((0, 700), (700, 2999), (2999, 3000)):
        rp2, c2, v2 = synth.generate_device(n, m, nnz, 0.8, 1.0, seed=11, user_range=(a, b))
        assert torch.equal(rp2, rp[a:b + 1] - rp[a])
        assert torch.equal(c2, c[rp[a]:rp[b]]) and torch.equal(v2, v[rp[a]:rp[b]])


def test_every_baseline_config_plans_its_nnz():
    # the degree plan of each BASELINE config sums to its nnz exactly (the fill then
    # makes every user reach its degree; full-size generation is checked on the GPU)
    for name, cfg in synth.CONFIGS.items():
        if cfg["n"] > 2_000_000:
            continue
        d = synth.degrees(cfg["n"], cfg["m"], cfg["nnz"], cfg["alpha_u"], cfg["seed"])
        assert int(d.sum()) == cfg["nnz"], name
        assert int(d.min()) >= 1 and int(d.max()) <= cfg["m"] // 2
