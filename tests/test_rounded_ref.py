"""CPU tier: tests/rounded_ref.py with the identity rounding hook IS the oracle (float64, 1e-12) -- which pins the rounded-operand
references of tests/test_gpu_fused_oracle.py to oracle/mimrl_ref.py -- and its bf16 hook rounds to nearest-even with a
straight-through gradient."""
import numpy as np
import torch

from oracle import mimrl_ref as R
from tests import rounded_ref as Q
from tests.helpers import case, oracle_params


def _p64(opt, seed):
    p = oracle_params(opt, seed, torch.float64)
    g = torch.Generator().manual_seed(7)
    return {k: (v + 0.02 * torch.randn(v.shape, generator=g, dtype=torch.float64) if k.endswith(".bias") else v) for k, v in p.items()}


def test_identity_hook_is_the_oracle_cube():
    for name in ("tiny_sep", "tiny_odd", "tiny_alt"):
        c, opt, batch, banks = case(name)
        p = _p64(opt, c["seed"])
        x = torch.randn(c["B"], opt.time_len, 3, 128, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
        want = R.cube_mlp(p, opt, x)
        for rt in ((False,) if opt.ln_first else (True, False)):
            got = Q.cube_mlp_q(p, opt, x, Q.EXACT, rt)
            assert torch.allclose(got, want, rtol=0, atol=1e-12), name


def test_identity_hook_is_the_oracle_critics():
    for name in ("tiny_sep", "tiny_cat"):
        c, opt, batch, banks = case(name)
        p = _p64(opt, c["seed"])
        feats = 0.3 * torch.randn(4, c["B"], 128, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
        mis, scores = Q.mi_terms_q(p, opt, feats, Q.EXACT)
        for n, m, s in zip(R.VMI_NAMES, mis, scores):
            ix, iy = Q.MI_WIRE[n]
            want = R.critic_scores(p, n, opt.critic_type, feats[ix], feats[iy])
            assert torch.allclose(s, want, rtol=0, atol=1e-12), (name, n)
            mi, _ = R.vmi_estimate(p, n, opt, feats[ix], feats[iy])
            assert abs(float(m) - float(mi)) < 1e-12


def test_identity_hook_is_the_oracle_cmi():
    c, opt, batch, banks = case("tiny_sep")
    p = _p64(opt, c["seed"])
    g = torch.Generator().manual_seed(4)
    n = 8
    x, y, z = (torch.randn(n, 128, dtype=torch.float64, generator=g) for _ in range(3))
    kx, ky, kz = (torch.randn(n, 128, dtype=torch.float64, generator=g) for _ in range(3))
    for name in R.VCMI_NAMES:
        cmi_o, bce_o = R.vcmi_estimate(p, name, opt, x, y, z, kx, ky, kz)
        batch_ = torch.cat([torch.cat([x, y, z], 1), torch.cat([kx, ky, kz], 1)], 0)
        _, bce, cmi = Q.cmi_terms_q(p, name, batch_, Q.EXACT)
        assert abs(float(cmi) - float(cmi_o)) < 1e-12 and abs(float(bce) - float(bce_o)) < 1e-12, name


def test_identity_rounding_has_the_oracles_gradients():
    """the custom backward of rounded_ref.mm (dx = g w, dw = g^T x) is autograd's when nothing is rounded"""
    c, opt, batch, banks = case("tiny_sep")
    p = _p64(opt, c["seed"])
    x = torch.randn(c["B"], opt.time_len, 3, 128, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    names = [n for n in p if n.startswith("mlp_encoder.")]
    res = []
    for fn in (lambda q, xx: R.cube_mlp(q, opt, xx), lambda q, xx: Q.cube_mlp_q(q, opt, xx, Q.EXACT)):
        leaves = {n: p[n].clone().requires_grad_(True) for n in names}
        xr = x.clone().requires_grad_(True)
        out = fn({**p, **leaves}, xr)
        res.append(torch.autograd.grad(out.square().sum(), [xr] + [leaves[n] for n in names]))
    for a, b in zip(*res):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-10)      # (values are O(10..1000))


def test_rounding_hooks():
    x = torch.tensor([1.0, 1.00390625, 1.005859375, -3.14159, 1e-30], dtype=torch.float64)
    y = Q.r_bf16(x)
    assert y.dtype == torch.float64
    np.testing.assert_array_equal(y.numpy()[:3], [1.0, 1.0, 1.0078125])      # ties to even, then up
    assert abs(float(y[3]) + 3.140625) < 1e-12
    np.testing.assert_array_equal(Q.r_f16(torch.tensor([1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11], dtype=torch.float64)).numpy(),
                                  [1.0, 1.0 + 2.0 ** -9])
    # a product: forward on fp16 operands, backward on bf16 ones
    xx = torch.tensor([[1.0 + 2.0 ** -10]], dtype=torch.float64, requires_grad=True)
    ww = torch.tensor([[3.0]], dtype=torch.float64, requires_grad=True)
    out = Q.mm(xx, ww, Q.F16_FWD)
    assert float(out) == 3.0 * (1.0 + 2.0 ** -10)
    out.backward()
    assert float(ww.grad) == 1.0 and float(xx.grad) == 3.0                   # bf16(1 + 2^-10) = 1


def test_identity_rounding_is_the_oracles_gru_and_its_autograd():
    """_GruDirQ writes the recurrence kernels' forward AND backward out by hand (the BPTT formulas of gru.hip, so that the rounding points of
    the backward kernel can be placed); with the rounding hooks set to the identity it must be oracle.gru_direction and torch autograd of it
    -- ragged lengths, both directions -- and encoders_q must be the first half of oracle.model_forward."""
    torch.manual_seed(0)
    B, T, D, H = 5, 7, 6, 128
    x = torch.randn(B, T, D, dtype=torch.float64, requires_grad=True)
    lens = torch.tensor([7, 3, 1, 5, 7])
    ws = [(torch.randn(3 * H, D, dtype=torch.float64) * 0.3).requires_grad_(True), (torch.randn(3 * H, H, dtype=torch.float64) * 0.1).requires_grad_(True),
          (torch.randn(3 * H, dtype=torch.float64) * 0.1).requires_grad_(True), (torch.randn(3 * H, dtype=torch.float64) * 0.1).requires_grad_(True)]
    do = torch.randn(B, T, H, dtype=torch.float64)
    for rev in (False, True):
        o1 = R.gru_direction(x, lens, ws[0], ws[1], ws[2], ws[3], rev)
        g1 = torch.autograd.grad((o1 * do).sum(), [x] + ws)
        o2 = Q._GruDirQ.apply(Q.mm(x, ws[0], Q.EXACT) + ws[2], ws[1], ws[3], lens, rev, Q.identity)
        g2 = torch.autograd.grad((o2 * do).sum(), [x] + ws)
        assert (o1 - o2).abs().max() < 1e-12
        for a, b in zip(g1, g2):
            assert (a - b).abs().max() < 1e-12
    c, opt, batch, banks = case("tiny_ragged", torch.float64)
    p = oracle_params(opt, c["seed"], torch.float64)
    xq, tf, af, vf = Q.encoders_q(p, opt, batch[0], batch[1], batch[2], Q.EXACT, Q.identity)
    pred, F_F, T_F, A_F, V_F = R.model_forward(p, opt, batch[0], batch[1], batch[2])
    for a, b in ((tf, T_F), (af, A_F), (vf, V_F)):
        assert (a - b).abs().max() < 1e-12
    assert (R.cube_mlp(p, opt, xq).mean(2).mean(1) - F_F).abs().max() < 1e-12
