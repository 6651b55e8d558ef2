"""-m gpu: operator-level parity of the HIP kernels (through the C ABI) against the CPU oracle."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from mimrl_amd import _lib
from oracle import mimrl_ref as R
from tests.gpu_helpers import P, assert_close, dev, grad_close, stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    lib = _lib.load()
    _lib.check(lib.mimrl_device_check())
    return lib


def _gemm(lib, A, B, Cm, M, N, K, batch, st, bias_n=None, bias_m=None, alpha=1.0, beta=0.0, act=0, prec=0):
    arr = (C.c_int64 * 9)(*st)
    _lib.check(lib.mimrl_op_gemm(stream(), P(A), P(B), P(Cm), M, N, K, batch, arr, P(bias_n), P(bias_m), alpha, beta, act, prec))
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (100, 70, 45), (1, 2, 384), (50, 384, 50), (333, 128, 768)])
def test_gemm_nt_bias_act(lib, M, N, K):
    g = np.random.default_rng(0)
    a, w, b = g.standard_normal((M, K)), g.standard_normal((N, K)), g.standard_normal(N)
    A, W, Bv = dev(a), dev(w), dev(b)
    out = torch.zeros(M, N, device="cuda")
    _gemm(lib, A, W, out, M, N, K, 1, (K, 1, 0, 1, K, 0, N, 1, 0), bias_n=Bv, act=1)
    ref = np.maximum(a @ w.T + b, 0)
    assert_close(out.cpu().numpy(), ref, 1e-5, 1e-4, "fp32 gemm")
    _gemm(lib, A, W, out, M, N, K, 1, (K, 1, 0, 1, K, 0, N, 1, 0), bias_n=Bv, act=1, prec=1)
    assert_close(out.cpu().numpy(), ref, 2e-2, 2e-2 * math.sqrt(K), "bf16 gemm")


def test_gemm_batched_left_multiply_and_tn(lib):
    """The CubeMLP L-axis mix: Y_b[m,C] = W[m,l] X_b[l,C] + bias_m, and a batch-reduced weight gradient."""
    g = np.random.default_rng(1)
    Bn, l, m, Cc = 5, 50, 10, 384
    w, x, bm = g.standard_normal((m, l)), g.standard_normal((Bn, l, Cc)), g.standard_normal(m)
    Wd, Xd, bd = dev(w), dev(x), dev(bm)
    Y = torch.zeros(Bn, m, Cc, device="cuda")
    _gemm(lib, Wd, Xd, Y, m, Cc, l, Bn, (l, 1, 0, Cc, 1, l * Cc, Cc, 1, m * Cc), bias_m=bd)
    ref = np.einsum("ml,blc->bmc", w, x) + bm[None, :, None]
    assert_close(Y.cpu().numpy(), ref, 1e-5, 1e-4, "left-multiply")
    # dW[m,l] = sum_b dY_b X_b^T via atomic accumulation over the batch (sc_b = 0)
    dW = torch.zeros(m, l, device="cuda")
    arr = (C.c_int64 * 9)(Cc, 1, m * Cc, 1, Cc, l * Cc, l, 1, 0)
    lib.mimrl_op_gemm(stream(), P(Y), P(Xd), P(dW), m, l, Cc, Bn, arr, None, None, C.c_float(1.0), C.c_float(1.0), 0, 0)
    # beta=1 with plain stores would race across the batch; the engine uses the atomic mode: emulate per batch
    dW.zero_()
    for b in range(Bn):
        arr = (C.c_int64 * 9)(Cc, 1, 0, 1, Cc, 0, l, 1, 0)
        _lib.check(lib.mimrl_op_gemm(stream(), P(Y[b]), P(Xd[b]), P(dW), m, l, Cc, 1, arr, None, None, 1.0, 1.0, 0, 0))
    torch.cuda.synchronize()
    ref = np.einsum("bmc,blc->ml", Y.cpu().numpy().astype(np.float64), x)
    assert_close(dW.cpu().numpy(), ref, 1e-4, 1e-2, "batch-reduced TN")


def _bf16_round(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


_LAYOUTS = {   # (A stored as, B stored as): which axis is contiguous
    "nt": ("mk", "nk"),      # forward products            (KC, KC)
    "nn": ("mk", "kn"),      # data gradients / left mixes (KC, RC)
    "tn": ("km", "kn"),      # weight gradients            (RC, RC)
}


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K,batch,atomic", [(256, 256, 96, 1, 0), (200, 72, 100, 3, 0), (128, 128, 4096, 1, 1), (1000, 384, 256, 2, 0),
                                                (64, 64, 32, 1, 0), (132, 260, 36, 1, 0), (50, 52, 384, 8, 1)])
def test_gemm_fast_path_layouts(lib, layout, M, N, K, batch, atomic):
    """The bf16 fast path (gemm_fast_kernel: 128x128 / 128x64 / 64x64 tiles, double-buffered LDS, transposing LDS reads for
    row-contiguous operands) on every operand-layout combination it serves, ragged M / N / K tiles, batch, split-K with
    atomics -- against a float64 product of the bf16-ROUNDED operands (what is left is fp32 accumulation order)."""
    g = np.random.default_rng(M * 7 + N * 3 + K)
    a = g.standard_normal((batch, M, K)).astype(np.float32)
    b = g.standard_normal((batch, K, N)).astype(np.float32)
    la, lb = _LAYOUTS[layout]
    A = dev(a if la == "mk" else a.transpose(0, 2, 1).copy())
    Bm = dev(b if lb == "kn" else b.transpose(0, 2, 1).copy())
    sa = (K, 1, M * K) if la == "mk" else (1, M, M * K)
    sb = (N, 1, K * N) if lb == "kn" else (1, K, K * N)
    red = atomic and batch > 1                                  # batch as an extra reduction axis (sc_b = 0)
    out = torch.zeros((1 if red else batch), M, N, device="cuda")
    st = (*sa, *sb, N, 1, 0 if red else M * N)
    arr = (C.c_int64 * 9)(*st)
    _lib.check(lib.mimrl_op_gemm(stream(), P(A), P(Bm), P(out), M, N, K, batch, arr, None, None, 1.0, 0.0, 256 * atomic, 1))
    torch.cuda.synchronize()
    ref = np.einsum("bmk,bkn->bmn", _bf16_round(a), _bf16_round(b))
    if red:
        ref = ref.sum(0, keepdims=True)
    err = np.abs(out.cpu().numpy() - ref).max()
    assert err <= 3e-6 * K * (batch if red else 1) + 1e-5, f"{layout} {M}x{N}x{K}: max |err| {err}"


def test_gemm_fast_path_dual_gap_epilogues(lib):
    """Second product (C = A.B + A2.B2: both directions of a bi-GRU data gradient), the row gap in A (dgh = columns
    [0,2H) u [3H,4H) of dg), act'(u) epilogue and fused column sums -- the parts of GemmDesc the engine drives."""
    g = np.random.default_rng(5)
    M, N, K = 640, 256, 384
    a1, a2 = g.standard_normal((M, 512)).astype(np.float32), g.standard_normal((M, 512)).astype(np.float32)
    w1, w2 = g.standard_normal((K, N)).astype(np.float32), g.standard_normal((K, N)).astype(np.float32)
    u = g.standard_normal((M, N)).astype(np.float32)
    out = torch.zeros(M, N, device="cuda")
    cs = torch.zeros(N, device="cuda")
    st = (C.c_int64 * 9)(512, 1, 0, N, 1, 0, N, 1, 0)
    st2 = (C.c_int64 * 6)(512, 1, 0, N, 1, 0)
    A1, A2, W1, W2, U = dev(a1), dev(a2), dev(w1), dev(w2), dev(u)
    _lib.check(lib.mimrl_op_gemm_ex(stream(), P(A1), P(W1), P(out), M, N, K, 1, st, P(A2), P(W2), K, st2, 0, 0, None, P(U), P(cs), 1, 1))
    torch.cuda.synchronize()
    ref = (_bf16_round(a1[:, :K]) @ _bf16_round(w1) + _bf16_round(a2[:, :K]) @ _bf16_round(w2)) * (u > 0)
    assert np.abs(out.cpu().numpy() - ref).max() <= 3e-6 * 2 * K + 1e-5
    assert_close(cs.cpu().numpy(), ref.sum(0), 1e-4, 2e-3, "fused column sums")
    # TN with a gap in A's row axis: rows m >= 256 of A^T live 128 further on (columns [0,256) u [384,512) of a1)
    hp = g.standard_normal((M, 128)).astype(np.float32)
    Hp = dev(hp)
    dw = torch.zeros(384, 128, device="cuda")
    st = (C.c_int64 * 9)(1, 512, 0, 128, 1, 0, 128, 1, 0)
    _lib.check(lib.mimrl_op_gemm_ex(stream(), P(A1), P(Hp), P(dw), 384, 128, M, 1, st, None, None, 0, None, 256, 128, None, None, None, 256, 1))
    torch.cuda.synchronize()
    sel = np.concatenate([a1[:, :256], a1[:, 384:512]], axis=1)
    ref = _bf16_round(sel).T @ _bf16_round(hp)
    assert np.abs(dw.cpu().numpy() - ref).max() <= 3e-6 * M + 1e-5


@pytest.mark.parametrize("kind", ["daxis", "laxis", "mixed"])
def test_gemm_wgrad_group_splitk(lib, kind):
    """The parked CubeMLP weight gradients as one grouped split-K launch: D-axis products dW = dY^T X (both operands
    row-contiguous, K = B*L*K rows) and the batch-reduced L-axis products dW += dY_b X_b^T (both k-contiguous, all
    samples into one output).  Ragged tiles, different K per problem; `mixed` = two layout classes -> the n-launch fallback."""
    g = np.random.default_rng(11)
    probs = []
    if kind in ("daxis", "mixed"):
        for (M, N, K) in [(128, 128, 9600), (128, 32, 9600), (32, 128, 9600), (20, 24, 1000), (64, 64, 96)]:
            a, b = g.standard_normal((K, M)).astype(np.float32), g.standard_normal((K, N)).astype(np.float32)
            probs.append((a, b, M, N, K, 1, (1, M, 0, N, 1, 0, N, 1, 0), _bf16_round(a).T @ _bf16_round(b)))
    if kind in ("laxis", "mixed"):
        for (M, N, K, nb) in [(50, 50, 384, 64), (50, 12, 384, 64), (12, 50, 384, 64), (8, 4, 128, 7)]:
            a, b = g.standard_normal((nb, M, K)).astype(np.float32), g.standard_normal((nb, N, K)).astype(np.float32)
            ref = np.einsum("bmk,bnk->mn", _bf16_round(a), _bf16_round(b))
            probs.append((a, b, M, N, K, nb, (K, 1, M * K, 1, K, N * K, N, 1, 0), ref))
    n = len(probs)
    As, Bs = [dev(q[0]) for q in probs], [dev(q[1]) for q in probs]
    Cs = [torch.zeros(q[2], q[3], device="cuda") for q in probs]
    pa = (C.c_void_p * n)(*[P(t) for t in As])
    pb = (C.c_void_p * n)(*[P(t) for t in Bs])
    pc = (C.c_void_p * n)(*[P(t) for t in Cs])
    dims = (C.c_int32 * (4 * n))(*[v for q in probs for v in (q[2], q[3], q[4], q[5])])
    st = (C.c_int64 * (9 * n))(*[v for q in probs for v in q[6]])
    _lib.check(lib.mimrl_op_gemm_wgrad_group(stream(), n, pa, pb, pc, dims, st, 1))
    torch.cuda.synchronize()
    for i, q in enumerate(probs):
        err = np.abs(Cs[i].cpu().numpy() - q[7]).max()
        assert err <= 3e-6 * q[4] * q[5] + 1e-5, f"{kind} problem {i} ({q[2]}x{q[3]}x{q[4]} batch {q[5]}): max |err| {err}"


@pytest.mark.parametrize("B", [32, 96, 128])
@pytest.mark.parametrize("tiled", [0, 1])
def test_mi_sep_infonce_fused(lib, B, tiled):
    """Separable critic + InfoNCE in one launch (scores = h g^T, bound, gradients of both tower outputs): the one-workgroup-per-
    estimator kernel and the row-tiled one (accumulating outputs) against float64 on the bf16-rounded tower outputs."""
    g = np.random.default_rng(B + tiled)
    E = 5
    tout = (0.5 * g.standard_normal((E, 2, B, 128))).astype(np.float32)
    gs = g.uniform(0.5, 1.5, E).astype(np.float32) * np.array([1, -1, 1, -1, 1], np.float32)
    T, GS = dev(tout), dev(gs)
    dT = torch.zeros(E, 2, B, 128, device="cuda")
    mi, ml = torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    _lib.check(lib.mimrl_op_mi_sep_infonce(stream(), P(T), P(dT), P(mi), P(ml), P(GS), E, B, tiled))
    torch.cuda.synchronize()
    tb = _bf16_round(tout).astype(np.float64)
    for e in range(E):
        gx, hy = tb[e, 0], tb[e, 1]
        S = hy @ gx.T
        lse = np.log(np.exp(S - S.max(1, keepdims=True)).sum(1)) + S.max(1)
        want = np.log(B) + np.mean(np.diag(S) - lse)
        assert abs(mi[e].item() - want) <= 1e-4 + 1e-4 * abs(want), (e, mi[e].item(), want)
        assert abs(ml[e].item() + want) <= 1e-4 + 1e-4 * abs(want)
        dS = gs[e] / B * (np.eye(B) - np.exp(S - lse[:, None]))
        # the kernels round dS (and the other operand) to bf16 for the two gradient products
        dh, dg = dS @ gx, dS.T @ hy
        got = dT[e].cpu().numpy()
        for name, got_m, want_m in (("d g(x)", got[0], dg), ("d h(y)", got[1], dh)):
            scale = np.abs(want_m).max() + 1e-12
            assert np.abs(got_m - want_m).max() <= 1.5e-2 * scale, (e, name, np.abs(got_m - want_m).max() / scale)


def _gru_case(B, T, d, seed, ragged):
    g = np.random.default_rng(seed)
    H = 128
    x = g.standard_normal((B, T, d)).astype(np.float32)
    lens = np.full(B, T, np.int32)
    if ragged:
        lens = g.integers(1, T + 1, size=B).astype(np.int32)
        lens[0] = T
        for b in range(B):
            x[b, lens[b]:] = 0
    k = 1 / math.sqrt(H)
    W = {n: g.uniform(-k, k, size=s).astype(np.float32) for n, s in
         [("wih_f", (384, d)), ("whh_f", (384, H)), ("bih_f", (384,)), ("bhh_f", (384,)),
          ("wih_r", (384, d)), ("whh_r", (384, H)), ("bih_r", (384,)), ("bhh_r", (384,))]}
    return x, lens, W


@pytest.mark.parametrize("B,T,ragged,prec", [(5, 7, False, 0), (16, 12, True, 0), (37, 9, True, 0), (16, 12, True, 1)])  # prec: bit0 -> bf16
def test_gru_layer_forward_backward(lib, B, T, ragged, prec):
    """One bidirectional layer: forward outputs and BPTT (dgx, dgh, h_prev) vs the oracle cell + autograd."""
    H, G = 128, 384
    x, lens, W = _gru_case(B, T, 20, 3, ragged)
    xt = torch.from_numpy(x).requires_grad_(True)
    Wt = {k: torch.from_numpy(v).requires_grad_(True) for k, v in W.items()}
    lt = torch.from_numpy(lens.astype(np.int64))
    of = R.gru_direction(xt, lt, Wt["wih_f"], Wt["whh_f"], Wt["bih_f"], Wt["bhh_f"], False)
    orr = R.gru_direction(xt, lt, Wt["wih_r"], Wt["whh_r"], Wt["bih_r"], Wt["bhh_r"], True)
    out_ref = torch.cat([of, orr], -1)
    gsel = torch.from_numpy(np.random.default_rng(9).standard_normal((B, T, 2 * H)).astype(np.float32))
    (out_ref * gsel).sum().backward()
    # device: hoisted input projections on the host (exact), recurrence on the GPU
    gx_f = dev(x.reshape(B * T, -1) @ W["wih_f"].T + W["bih_f"]).reshape(B, T, G).contiguous()
    gx_r = dev(x.reshape(B * T, -1) @ W["wih_r"].T + W["bih_r"]).reshape(B, T, G).contiguous()
    nsv = lib.mimrl_op_gru_saved_floats(B, T)
    sv_f, sv_r = torch.zeros(nsv, device="cuda"), torch.zeros(nsv, device="cuda")
    out = torch.full((B, T, 2 * H), float("nan"), device="cuda")
    lens_d = torch.from_numpy(lens).cuda()
    Wd = {k: dev(v) for k, v in W.items()}
    _lib.check(lib.mimrl_op_gru_forward(stream(), P(gx_f), P(gx_r), P(Wd["whh_f"]), P(Wd["whh_r"]), P(Wd["bhh_f"]),
                                        P(Wd["bhh_r"]), P(lens_d), P(out), P(sv_f), P(sv_r), B, T, prec))
    torch.cuda.synchronize()
    tol = 1e-5 if prec == 0 else 2e-2
    assert_close(out.cpu().numpy(), out_ref.detach().numpy(), tol, tol, "gru forward")
    dout = gsel.cuda()
    dg_f, dg_r = (torch.full((B, T, 4 * H), float("nan"), device="cuda") for _ in range(2))
    hp_f, hp_r = (torch.full((B, T, H), float("nan"), device="cuda") for _ in range(2))
    _lib.check(lib.mimrl_op_gru_backward(stream(), P(Wd["whh_f"]), P(Wd["whh_r"]), P(sv_f), P(sv_r), P(lens_d), P(out),
                                         P(dout), P(dg_f), P(dg_r), P(hp_f), P(hp_r), B, T, prec))
    torch.cuda.synchronize()
    rel = 2e-4 if prec == 0 else 4e-2
    dgx = {}
    for tag, dg, hp in (("f", dg_f, hp_f), ("r", dg_r, hp_r)):
        dg_n, hp_n = dg.cpu().numpy().astype(np.float64), hp.cpu().numpy().astype(np.float64)
        assert np.isfinite(dg_n).all() and np.isfinite(hp_n).all()
        dgx_n = dg_n[..., :G]                                           # [dr'|dz'|dn']
        dgh_n = np.concatenate([dg_n[..., :2 * H], dg_n[..., 3 * H:]], -1)   # [dr'|dz'|dn'*r]
        dgx[tag] = dgx_n
        x64 = x.reshape(B * T, -1).astype(np.float64)
        grad_close(dgx_n.reshape(B * T, G).T @ x64, Wt["wih_" + tag].grad.numpy(), rel, "dW_ih " + tag)
        grad_close(dgx_n.reshape(B * T, G).sum(0), Wt["bih_" + tag].grad.numpy(), rel, "db_ih " + tag)
        grad_close(dgh_n.reshape(B * T, G).T @ hp_n.reshape(B * T, H), Wt["whh_" + tag].grad.numpy(), rel, "dW_hh " + tag)
        grad_close(dgh_n.reshape(B * T, G).sum(0), Wt["bhh_" + tag].grad.numpy(), rel, "db_hh " + tag)
    dx = dgx["f"].reshape(B * T, G) @ W["wih_f"] + dgx["r"].reshape(B * T, G) @ W["wih_r"]
    grad_close(dx.reshape(B, T, -1), xt.grad.numpy(), rel, "dx")


@pytest.mark.parametrize("bound", [b for b in _lib.BOUNDS if b != "mine"])
@pytest.mark.parametrize("B", [8, 32, 128])
def test_mi_bounds_value_and_gradient(lib, bound, B):
    g = np.random.default_rng(4)
    E = 3
    s = (g.standard_normal((E, B, B)) * 1.5).astype(np.float32)
    gs = np.array([-1.0, 0.5, -0.01], np.float32)
    S, dS, mi, GS = dev(s), torch.zeros(E, B, B, device="cuda"), torch.zeros(E, device="cuda"), dev(gs)
    _lib.check(lib.mimrl_op_mi_bound(stream(), P(S), P(dS), P(mi), P(GS), E, B, _lib.BOUNDS[bound]))
    torch.cuda.synchronize()
    for e in range(E):
        st = torch.from_numpy(s[e]).double().requires_grad_(True)
        val = R.BOUNDS[bound](st)
        (val * float(gs[e])).backward()
        assert_close(mi[e].item(), val.item(), 1e-4, 2e-5, f"{bound} value")
        grad_close(dS[e].cpu().numpy(), st.grad.numpy(), 1e-3, f"{bound} gradient")


@pytest.mark.parametrize("bound", ["tuba", "interpolate"])
@pytest.mark.parametrize("B", [8, 32, 128])
def test_bounds_with_log_baseline(lib, bound, B):
    """tuba / interpolate with a per-row log-baseline (VMI.py:72-110): value, d/dscores and d/dbaseline vs autograd."""
    g = np.random.default_rng(6)
    E = 3
    s = (g.standard_normal((E, B, B)) * 1.2).astype(np.float32)
    lbv = (g.standard_normal((E, B)) * 0.7 - 0.3).astype(np.float32)
    gs = np.array([-1.0, 0.5, -0.01], np.float32)
    S, dS, GS = dev(s), torch.zeros(E, B, B, device="cuda"), dev(gs)
    LB, dLB, mi = dev(lbv), torch.zeros(E, B, device="cuda"), torch.zeros(E, device="cuda")
    _lib.check(lib.mimrl_op_mi_bound_baseline(stream(), P(S), P(dS), P(mi), P(GS), P(LB), P(dLB), E, B, _lib.BOUNDS[bound]))
    torch.cuda.synchronize()
    for e in range(E):
        st = torch.from_numpy(s[e]).double().requires_grad_(True)
        lt = torch.from_numpy(lbv[e]).double().reshape(B, 1).requires_grad_(True)
        val = R.BOUNDS[bound](st, lt)
        (val * float(gs[e])).backward()
        assert_close(mi[e].item(), val.item(), 1e-4, 2e-5, f"{bound} value")
        grad_close(dS[e].cpu().numpy(), st.grad.numpy(), 1e-3, f"{bound} d/dscores")
        grad_close(dLB[e].cpu().numpy(), lt.grad.numpy().reshape(-1), 1e-3, f"{bound} d/dbaseline")


@pytest.mark.parametrize("B", [8, 64])
def test_mine_bound_loss_term_and_both_gradient_forms(lib, B):
    """`mine` (Model.py:121-125): value = dv form; loss term = mean(t) - mean(et)/ma_et (not negated).  Estimators flagged in
    `lossform` get the gradient of coefficient * loss term, the others that of -coefficient * value (Model.py:386)."""
    g = np.random.default_rng(5)
    E = 3
    s = (g.standard_normal((E, B, B)) * 1.2).astype(np.float32)
    coef = np.array([1.0, 0.5, 0.01], np.float32)
    S, dS, GS = dev(s), torch.zeros(E, B, B, device="cuda"), dev(-coef)
    mi, ml = torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    _lib.check(lib.mimrl_op_mi_bound_ex(stream(), P(S), P(dS), P(mi), P(ml), P(GS), E, B, _lib.BOUNDS["mine"], 0b101))
    torch.cuda.synchronize()
    for e in range(E):
        st = torch.from_numpy(s[e]).double().requires_grad_(True)
        t = st.diag()
        et = torch.exp(st) * (1.0 - torch.eye(B, dtype=st.dtype))
        val = t.mean() - R._logmeanexp_nodiag(st)
        loss = t.mean() - (1 / (0.99 + 0.01 * et.mean())).detach() * et.mean()
        lossform = bool((0b101 >> e) & 1)
        (float(coef[e]) * (loss if lossform else -val)).backward()
        assert_close(mi[e].item(), val.item(), 1e-4, 2e-5, "mine value")
        assert_close(ml[e].item(), (loss if lossform else -val).item(), 1e-4, 2e-5, "mine loss term")
        grad_close(dS[e].cpu().numpy(), st.grad.numpy(), 1e-3, f"mine gradient (lossform={lossform})")


def _knn_bank(kind, g, N, dz):
    """Banks for the kNN tests.  `dup`: every row occurs three times (exact ties in both the expansion score and the exact distance: the
    MFMA filter cannot prove itself complete and the kernel must take its exact-scan fallback); `collapsed`: rows = one vector + 1e-4
    noise (feature collapse: distances ~1e-6 of the norms, below the rounding bound of the norm expansion -> fallback as well);
    `offset`: a large common mean (|z|^2 >> distances: the regime where the expansion loses most bits)."""
    if dz == 1:
        return g.uniform(-3, 3, size=(N, 1)).astype(np.float32)
    Z = g.standard_normal((N, dz)).astype(np.float32)
    if kind == "dup":
        Z = Z[np.arange(N) % ((N + 2) // 3)]
    elif kind == "collapsed":
        Z = (Z[:1] + 1e-4 * Z).astype(np.float32)
    elif kind == "offset":
        Z = (Z + 30.0).astype(np.float32)
    return np.ascontiguousarray(Z)


KNN_CASES = [(128, 300, 16, 2, "normal"), (1, 300, 16, 2, "normal"), (128, 1284, 64, 2, "normal"), (1, 1284, 64, 3, "normal"),
             (128, 70, 33, 4, "normal"),
             # round 4 (fp32 MFMA tiles + exact refinement, knn_mfma.hip): cfg3's shape, two anchor blocks, one / three / seven N-tiles,
             # k = 3..4 (six survivors), k = 5 (exact scan), a generic width, and the banks that defeat the filter
             (128, 16326, 128, 2, "normal"), (128, 5000, 256, 2, "normal"), (128, 1284, 40, 2, "normal"), (128, 3000, 100, 4, "normal"),
             (128, 3000, 100, 3, "offset"), (128, 2000, 64, 5, "normal"), (64, 500, 32, 2, "normal"), (1, 16326, 128, 2, "normal"),
             (128, 1284, 64, 2, "dup"), (128, 900, 32, 4, "dup"), (128, 1284, 64, 2, "collapsed"), (128, 16326, 128, 2, "offset")]


@pytest.mark.parametrize("dz,N,m,k,kind", KNN_CASES)
def test_knn_matches_exact_bruteforce(lib, dz, N, m, k, kind):
    g = np.random.default_rng(5)
    Z = _knn_bank(kind, g, N, dz)
    anchors = g.choice(N, size=m, replace=False).astype(np.int32)
    out = torch.full((m, k), -1, dtype=torch.int32, device="cuda")
    Zd, Ad = dev(Z), torch.from_numpy(anchors).cuda()      # keep both alive across the asynchronous launch
    _lib.check(lib.mimrl_op_knn(stream(), P(Zd), dz, N, P(Ad), m, k, P(out)))
    torch.cuda.synchronize()
    ref = R.knn_indices(Z, anchors.astype(np.int64), k)
    got = out.cpu().numpy()
    if not np.array_equal(got, ref):   # only exact ties (in fp32) may legitimately differ
        Z64 = Z.astype(np.float64)
        for i, j in zip(*np.nonzero(got != ref)):
            dg = ((Z64[got[i, j]] - Z64[anchors[i]]) ** 2).sum()
            dr = ((Z64[ref[i, j]] - Z64[anchors[i]]) ** 2).sum()
            assert abs(dg - dr) <= 1e-6 * dr + 1e-30, f"anchor {i} slot {j}: got row {got[i, j]} (d2={dg}) want {ref[i, j]} (d2={dr})"
            assert got[i, j] not in anchors
    assert got.min() >= 0 and got.max() < N and not np.isin(got, anchors).any()
    assert all(len(set(r)) == k for r in got.tolist())          # k DISTINCT neighbours per anchor
    if kind == "dup":                                            # exact ties resolve to the lower row, as in the brute force / np.argsort(stable)
        Z64 = Z.astype(np.float64)
        for i in range(m):
            d = ((Z64[got[i]] - Z64[anchors[i]]) ** 2).sum(1)
            assert np.all(np.diff(d) >= 0)
            for j in range(k - 1):
                if d[j] == d[j + 1]:
                    assert got[i, j] < got[i, j + 1]


def _mix32(x):
    M = np.uint64(0xFFFFFFFF)
    x = x & M
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M
    x ^= x >> np.uint64(16)
    return x


def _anchor_keys(N, seed, st, stream_id, c):
    """The key of every bank row, restated from csrc/knn_mfma.hip::anchor_hash (host-side numpy; the device draw = the m smallest keys)."""
    M = np.uint64(0xFFFFFFFF)
    rows = np.arange(N, dtype=np.uint64)
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    salt = _mix32(np.uint64((st * 0x9E3779B9 + stream_id + 977 * c) & 0xFFFFFFFF))
    h = _mix32(rows ^ salt ^ lo)
    h = _mix32((h + ((hi * np.uint64(0x85ebca6b)) & M) + np.uint64(0x632be5ab)) & M)
    return (h << np.uint64(32)) | rows


@pytest.mark.parametrize("N,m", [(1284, 64), (16326, 128), (70, 33), (64, 64), (40000, 128), (163, 16), (1284, 1), (100000, 512)])
def test_sample_anchors_is_the_m_smallest_keys(lib, N, m):
    """Model.py:81 on the device (the mode bench.py times): per call, m DISTINCT rows in [0, N) = exactly the m smallest (hash, row) keys in
    key order (the definition the round-1 bitonic sort implemented), different per call / step / stream id -- any bank size."""
    seed, ncall = 0x1234567811223344, 6
    step = torch.tensor([41], dtype=torch.int32, device="cuda")
    out = torch.full((ncall, m), -1, dtype=torch.int32, device="cuda")
    seen = set()
    for stream_id, step_add in ((101, 0), (102, 0), (101, 1)):
        _lib.check(lib.mimrl_op_sample_anchors(stream(), P(out), ncall, m, N, seed, P(step), stream_id, step_add))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for c in range(ncall):
            keys = _anchor_keys(N, seed, 41 + step_add, stream_id, c)
            want = (np.sort(keys)[:m] & np.uint64(0xFFFFFFFF)).astype(np.int64)
            assert np.array_equal(got[c], want), (stream_id, step_add, c)
            assert len(set(got[c].tolist())) == m and got[c].min() >= 0 and got[c].max() < N
            if m >= 16 and N >= 4 * m:
                seen.add(tuple(got[c].tolist()))
    if m >= 16 and N >= 4 * m:
        assert len(seen) == 3 * ncall                           # every (call, step, stage) draws its own subset


def test_sample_anchors_inclusion_is_uniform(lib):
    """400 steps x 6 calls of m = 20 out of N = 200: every row is drawn with frequency m / N (binomial 5 sigma), and the position of a row in
    the draw is uniform too (first-slot frequencies)."""
    N, m, ncall, steps = 200, 20, 6, 400
    step = torch.tensor([0], dtype=torch.int32, device="cuda")
    out = torch.empty(steps, ncall, m, dtype=torch.int32, device="cuda")
    for s in range(steps):
        _lib.check(lib.mimrl_op_sample_anchors(stream(), P(out[s]), ncall, m, N, 7, P(step), 101, s))
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(-1, m)
    n = got.shape[0]
    cnt = np.bincount(got.reshape(-1), minlength=N)
    p = m / N
    assert np.abs(cnt - n * p).max() <= 5 * np.sqrt(n * p * (1 - p)), (cnt.min(), cnt.max(), n * p)
    first = np.bincount(got[:, 0], minlength=N)
    assert np.abs(first - n / N).max() <= 5 * np.sqrt(n / N) + 1


@pytest.mark.parametrize("hardtanh", [0, 1])
def test_cmi_loss_and_gradient(lib, hardtanh):
    g = np.random.default_rng(6)
    E, n = 2, 12
    logits = (g.standard_normal((E, 2 * n, 2)) * (0.4 if hardtanh else 4.0)).astype(np.float32)
    if hardtanh:
        logits += 0.5
    logits[0, 0, 0] = 11.0      # exercises the +-10 clamp (Model.py:69)
    gb, gc = np.array([1.0, 0.3], np.float32), np.array([-0.01, 0.7], np.float32)
    Ld, dL = dev(logits), torch.zeros(E, 2 * n, 2, device="cuda")
    bce, cmi = torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    gbd, gcd = dev(gb), dev(gc)
    _lib.check(lib.mimrl_op_cmi_loss(stream(), P(Ld), P(dL), P(bce), P(cmi), P(gbd), P(gcd), E, n, hardtanh))
    torch.cuda.synchronize()
    for e in range(E):
        lt = torch.from_numpy(logits[e]).double().requires_grad_(True)
        o = torch.clamp(lt, -10, 10)
        gam = torch.nn.functional.hardtanh(o, 1e-4, 1 - 1e-4) if hardtanh else torch.sigmoid(o)
        tgt = torch.zeros(2 * n, 2, dtype=torch.float64)
        tgt[:n, 0] = 1
        tgt[n:, 1] = 1
        b = torch.nn.functional.binary_cross_entropy(gam, tgt)
        lr = torch.log(gam[:, 0] / (1 - gam[:, 0] + 1e-6))
        cm = 1 + lr[:n].sum() / (2 * n) - lr[n:].sum() / (2 * n)
        (float(gb[e]) * b + float(gc[e]) * cm).backward()
        assert_close(bce[e].item(), b.item(), 1e-4, 1e-6, "bce")
        assert_close(cmi[e].item(), cm.item(), 1e-4, 1e-5, "cmi")
        grad_close(dL[e].cpu().numpy(), lt.grad.numpy(), 1e-3, "dlogits")


def test_fused_clip_adam_matches_torch_semantics(lib):
    g = np.random.default_rng(7)
    n = 5000
    p0 = g.standard_normal(n).astype(np.float32)
    p, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    lr, step = dev(np.array([4e-3], np.float32)), torch.zeros(1, dtype=torch.int32, device="cuda")
    ref = {"w": torch.from_numpy(p0.copy())}
    adam = R.AdamState(ref, ["w"])
    for it in range(5):
        gr = (g.standard_normal(n) * 2).astype(np.float32)
        step += 1
        grd = dev(gr)
        _lib.check(lib.mimrl_op_adam(stream(), P(p), P(grd), P(m), P(v), n, P(lr), P(step), 0.9, 0.999, 1e-8, 0.01, 1.5))
        torch.cuda.synchronize()
        adam.step(ref, {"w": torch.from_numpy(gr).clamp(-1.5, 1.5)}, 4e-3, 0.01)
    torch.cuda.synchronize()
    assert_close(p.cpu().numpy(), ref["w"].numpy(), 1e-5, 1e-6, "adam params")


@pytest.mark.parametrize("nb,rows,brows,dims", [(10, 128, 128, [128, 256, 256, 256, 128]), (6, 200, 256, [384, 256, 256, 256, 2]),
                                                 (3, 37, 64, [128, 256, 2]), (2, 32, 32, [64, 128])])
def test_fused_mlp_stack_forward_backward(lib, nb, rows, brows, dims):
    """mlp_fused.hip (critic towers VMI.py:13-22 / CMI classifier Model.py:47-72 as ONE kernel per direction) against a
    torch fp32 reference that rounds the MFMA operands to bf16 the same way (so the comparison is tight)."""
    torch.manual_seed(0)
    nl = len(dims) - 1
    pstride = sum(dims[l + 1] * dims[l] + dims[l + 1] for l in range(nl)) + 64
    pstride += (-pstride) % 64
    flat = torch.zeros(nb * pstride, device="cuda")
    gflat = torch.zeros_like(flat)
    offs, o = [], 0
    for l in range(nl):
        offs.append((o, o + dims[l + 1] * dims[l]))
        o += dims[l + 1] * dims[l] + dims[l + 1]
        o += (-o) % 4
    Ws, bs = [], []
    for g in range(nb):
        for l in range(nl):
            w = torch.randn(dims[l + 1], dims[l], device="cuda") / math.sqrt(dims[l])
            b = torch.randn(dims[l + 1], device="cuda") * 0.1
            flat[g * pstride + offs[l][0]: g * pstride + offs[l][0] + w.numel()] = w.flatten()
            flat[g * pstride + offs[l][1]: g * pstride + offs[l][1] + b.numel()] = b
            Ws.append(w); bs.append(b)
    x = torch.randn(nb, brows, dims[0], device="cuda")
    acts = [torch.zeros(nb, brows, dims[l + 1], device="cuda") for l in range(nl - 1)]
    out = torch.zeros(nb, brows, dims[nl], device="cuda")
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])
    cdims = (C.c_int32 * (nl + 1))(*dims)
    Wp = [flat[offs[l][0]:] for l in range(nl)]
    bp = [flat[offs[l][1]:] for l in range(nl)]
    _lib.check(lib.mimrl_op_mlp_stack_forward(stream(), nb, rows, brows, nl, cdims, arr(Wp), arr(bp), C.c_int64(pstride), P(x),
                                              arr(acts + [None]), P(out)))
    torch.cuda.synchronize()
    r16 = lambda t: t.to(torch.bfloat16).float()
    # reference (per group), bf16-rounded operands, fp32 accumulate
    dout = torch.randn(nb, brows, dims[nl], device="cuda")
    dzs = [None] + [torch.zeros(nb, brows, dims[l], device="cuda") for l in range(1, nl)]
    din = torch.zeros(nb, brows, dims[0], device="cuda")
    dbp = [gflat[offs[l][1]:] for l in range(nl - 1)]
    _lib.check(lib.mimrl_op_mlp_stack_backward(stream(), nb, rows, brows, nl, cdims, arr(Wp), C.c_int64(pstride), arr(acts + [None]),
                                               P(dout), arr(dzs + [None]), P(din), arr(dbp + [None])))
    torch.cuda.synchronize()
    for g in range(nb):
        a = x[g, :rows]
        ref_acts = []
        for l in range(nl):
            z = r16(a) @ r16(Ws[g * nl + l]).T + bs[g * nl + l]
            a = torch.relu(z) if l < nl - 1 else z
            ref_acts.append(a)
        for l in range(nl - 1):
            assert_close(acts[l][g, :rows].cpu().numpy(), ref_acts[l].cpu().numpy(), 1e-2, 5e-3, f"g{g} act{l}")
        assert_close(out[g, :rows].cpu().numpy(), ref_acts[-1].cpu().numpy(), 1e-2, 5e-3, f"g{g} out")
        dz = dout[g, :rows]
        for l in range(nl - 1, -1, -1):
            d = r16(dz) @ r16(Ws[g * nl + l])
            if l > 0:
                d = d * (acts[l - 1][g, :rows] > 0)
                assert_close(dzs[l][g, :rows].cpu().numpy(), d.cpu().numpy(), 1e-2, 1e-2, f"g{g} dz{l}")
                got_db = gflat[g * pstride + offs[l - 1][1]: g * pstride + offs[l - 1][1] + dims[l]]
                assert_close(got_db.cpu().numpy(), dzs[l][g, :rows].sum(0).cpu().numpy(), 1e-3, 1e-3, f"g{g} db{l - 1}")
                dz = dzs[l][g, :rows]
            else:
                assert_close(din[g, :rows].cpu().numpy(), d.cpu().numpy(), 1e-2, 1e-2, f"g{g} din")
        if rows < brows:
            assert float(out[g, rows:].abs().max()) == 0.0 and float(din[g, rows:].abs().max()) == 0.0


def _gemm16(lib, A, B, Cm, M, N, K, batch, st, flags, A2=None, B2=None, K2=0, st2=None, batch_in=0, st_bo=None, bias=None):
    arr = (C.c_int64 * 9)(*st)
    arr2 = (C.c_int64 * 6)(*st2) if st2 else None
    arrb = (C.c_int64 * 5)(*st_bo) if st_bo else None
    _lib.check(lib.mimrl_op_gemm16(stream(), P(A), P(B), P(Cm), M, N, K, batch, arr, P(A2), P(B2), K2, arr2, batch_in, arrb, P(bias), flags))
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,f16,cf16", [(16384 + 100, 384, 256, True, False), (16384 + 100, 384, 256, True, True), (20000, 160, 64, False, False),
                                            (16384, 128, 192, False, True), (16640, 36, 64, True, False), (70000, 256, 128, False, False)])
def test_gemm_tall_projection_shape(lib, M, N, K, f16, cf16):
    """csrc/gemm_tall.hip (round 5): the LDS-DMA kernel gemm() picks for tall products with both operands stored in 16 bits and
    k-contiguous -- here the shape class of the GRU layer-1 input projection (Model.py:254-255): batch = (modality, direction) with the
    directions sharing A, bias per output column, fp32 or fp16 output, ragged last row block, N not a multiple of the 128-column tile,
    K = one k-tile.  Reference: float64 product of the SAME 16-bit operands (only the fp32 accumulation order differs)."""
    g = np.random.default_rng(M + N + K)
    t16 = torch.float16 if f16 else torch.bfloat16
    a = torch.from_numpy(g.standard_normal((2, M, K)).astype(np.float32)).to(t16).cuda()            # [modality][M, K]
    w = torch.from_numpy(g.standard_normal((2, 2, N, K)).astype(np.float32) * 0.1).to(t16).cuda()   # [modality][direction][N, K]
    b = dev(g.standard_normal((2, 2, N)))
    out = torch.zeros(2, 2, M, N, device="cuda", dtype=torch.float16 if cf16 else torch.float32)
    flags = 3 | (4 if f16 else 0) | (8 if cf16 else 0)
    _gemm16(lib, a, w, out, M, N, K, 4, (K, 1, 0, 1, K, N * K, N, 1, M * N), flags, batch_in=2,
            st_bo=(M * K, 2 * N * K, 2 * M * N, 2 * N, N), bias=b)
    ref = torch.einsum("mrk,mdnk->mdrn", a.double(), w.double()) + b.double()[:, :, None, :]
    got = out.double()
    tol = 2e-6 * K + 1e-5 + (2e-3 * ref.abs().max().item() if cf16 else 0.0)
    err = (got - ref).abs().max().item()
    assert err <= tol, f"tall {M}x{N}x{K} f16={f16} cf16={cf16}: max |err| {err} > {tol}"


def test_gemm_tall_two_segments(lib):
    """The dh0 shape class (autograd of Model.py:254-255): C = A1 . W1^T + A2 . W2^T with A1 / A2 = the first 384 of 512 columns of two
    bf16 row arrays (the two directions' dg rows) and W = one [N, 768] image holding both directions' k ranges, batch = modality."""
    g = np.random.default_rng(11)
    M, N, K = 16384 + 77, 256, 384
    a = torch.from_numpy(g.standard_normal((2, 2, M, 512)).astype(np.float32)).to(torch.bfloat16).cuda()   # [modality][direction][M, 512]
    w = torch.from_numpy(g.standard_normal((2, N, 2 * K)).astype(np.float32) * 0.1).to(torch.bfloat16).cuda()
    out = torch.full((2, M, N), 7.0, device="cuda")
    _gemm16(lib, a[:, 0], w, out, M, N, K, 2, (512, 1, 2 * M * 512, 1, 2 * K, N * 2 * K, N, 1, M * N), 3,
            A2=a[:, 1], B2=w[:, :, K:], K2=K, st2=(512, 1, 2 * M * 512, 1, 2 * K, N * 2 * K))
    ref = torch.einsum("mdrk,mndk->mrn", a[..., :K].double(), w.double().reshape(2, N, 2, K))
    err = (out.double() - ref).abs().max().item()
    assert err <= 2e-6 * 2 * K + 1e-5, f"two-segment tall product: max |err| {err}"


@pytest.mark.parametrize("M,N,K,gap,shared_b", [(384, 128, 20000 + 17, True, False), (384, 256, 16384 + 32 * 5, False, True), (384, 80, 16400, False, True),
                                                 (256, 256, 16384, False, False)])
def test_gemm_tall_weight_gradient_shape(lib, M, N, K, gap, shared_b, monkeypatch):
    """The recurrence weight-gradient shapes on the split-K fast path gemm() gives them (round 5 also had an opt-in LDS-DMA kernel for them that
    tied it; removed in round 6): C[M, N] += A^T B over K = B*T rows with A [K, 512] / B [K, N] stored bf16 and
    row-contiguous -- dW_hh = dgh^T h_prev (A rows [0, 256) u [384, 512) of dg: the gap) and dW_ih = dgx^T x (both directions share B) of
    Model.py:254-255's autograd; batch = (modality, direction); K not a multiple of the 32-row k-step (zero page), N = 80 (the packed
    layer-0 inputs).  Reference: float64 product of the same bf16 operands; the output is accumulated with float atomics over the k-split."""
    g = np.random.default_rng(M + N + K)
    a = torch.from_numpy(g.standard_normal((2, 2, K, 512)).astype(np.float32)).to(torch.bfloat16).cuda()       # [modality][direction][K, 512]
    nbm = 1 if shared_b else 2
    b = torch.from_numpy(g.standard_normal((2, nbm, K, N)).astype(np.float32) * 0.1).to(torch.bfloat16).cuda()
    out = torch.zeros(2, 2, M, N, device="cuda")
    flags = 3 | 16 | ((256 << 8) | (128 << 20) if gap else 0)
    _gemm16(lib, a, b, out, M, N, K, 4, (1, 512, K * 512, N, 1, 0 if shared_b else K * N, N, 1, M * N), flags, batch_in=2,
            st_bo=(2 * K * 512, nbm * K * N, 2 * M * N, 0, 0))
    cols = np.r_[0:256, 384:512] if gap else np.r_[0:M]
    ad = a.double()[..., cols]
    bd = b.double().expand(2, 2, K, N) if shared_b else b.double()
    ref = torch.einsum("mdkr,mdkn->mdrn", ad, bd)
    err = (out.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 3e-6 * scale + 1e-4, f"tall TN {M}x{N}x{K}: max |err| {err} (scale {scale})"


@pytest.mark.parametrize("rows,kp", [(48, 80), (6400, 80), (20000 + 17, 80), (4096 + 5, 40), (1600, 96), (33, 8)])
def test_gru_wgrad_one_pass_over_dg(lib, rows, kp):
    """gru_wgrad.hip (round 6): both layer-0 weight-gradient products of all four (modality, direction) sequences in ONE launch that reads
    dg once -- dW_ih[s] [384, kp] += dgx_s^T x_s (dgx = dg columns [0, 384)), dW_hh[s] [384, 128] += dgh_s^T h_prev_s (dgh = columns
    [0, 256) u [384, 512)), Model.py:254-255's autograd; x shared by the two directions of a modality; every operand stored as bf16.
    Row counts that are not a multiple of the 32-row k-tile or of the ring depth, fewer k-tiles than k-ranges, every packed width class
    (kp = 8, 40, 80, 96).  Reference: float64 product of the same bf16 operands; outputs accumulate with float atomics over the k-split, and
    on top of what the arrays held before."""
    g = np.random.default_rng(rows + kp)
    dg = torch.from_numpy(g.standard_normal((4, rows, 512)).astype(np.float32)).to(torch.bfloat16).cuda()
    x = torch.from_numpy(g.standard_normal((2, rows, kp)).astype(np.float32) * 0.5).to(torch.bfloat16).cuda()
    hp = torch.from_numpy(g.standard_normal((4, rows, 128)).astype(np.float32) * 0.3).to(torch.bfloat16).cuda()
    dwih = torch.full((4, 384, kp), 0.25, device="cuda")
    dwhh = torch.full((4, 384, 128), -0.5, device="cuda")
    arr = lambda ts: (C.c_void_p * 4)(*[t.data_ptr() for t in ts])
    _lib.check(lib.mimrl_op_gru_wgrad(stream(), arr([dg[s] for s in range(4)]), arr([x[s // 2] for s in range(4)]), arr([hp[s] for s in range(4)]),
                                      arr([dwih[s] for s in range(4)]), arr([dwhh[s] for s in range(4)]), rows, kp))
    torch.cuda.synchronize()
    d = dg.double()
    ref_ih = torch.einsum("skr,skn->srn", d[..., :384], x.double()[[0, 0, 1, 1]]) + 0.25
    ref_hh = torch.einsum("skr,skn->srn", d[..., np.r_[0:256, 384:512]], hp.double()) - 0.5
    for name, got, ref in (("dW_ih", dwih, ref_ih), ("dW_hh", dwhh, ref_hh)):
        err = (got.double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        assert err <= 3e-6 * scale + 1e-4, f"{name} rows={rows} kp={kp}: max |err| {err} (scale {scale})"


@pytest.mark.parametrize("E,rows,two,head,gen", [(3, 4096, True, True, False), (2, 100, True, True, True), (5, 16384, True, True, True), (1, 33, False, False, False),
                                                  (2, 65536, True, True, True), (4, 1000, False, True, False), (3, 2050, False, False, True)])
def test_concat_dw_whole_output_per_workgroup(lib, E, rows, two, head, gen):
    """concat_dw.hip (round 6): the stage-1 weight gradients of the concat critic's hidden layers (VMI.py:58-65 under autograd),
    dW_l[e] [256, 256] += dZ_l[e]^T A_{l-1}[e] over the B*B pair rows, both layers in one launch whose workgroups hold a whole 256 x 256
    output, operands stored as bf16 -- on extra workgroups of the same launch the score head's dw3[e] [256] += ds[e]^T a2[e] with a2
    stored as fp16 -- and (gen) dZ2 regenerated inside the kernel from ds, the score head's weight and the layer-2 sign words instead of
    read.  Row counts that are not a multiple of the 32-row k-tile, of the ring depth or of the score head's 128-row pass, fewer k-tiles
    than k-ranges, one layer only, with and without the score head, estimator outputs a stride apart.  Reference: float64 product of the
    same 16-bit operands; the outputs accumulate (float atomics over the k-split) on top of what they held."""
    g = np.random.default_rng(E * 1000 + rows)
    mk = lambda sc: torch.from_numpy(g.standard_normal((E, rows, 256)).astype(np.float32) * sc).to(torch.bfloat16).cuda()
    dz2, a1, dz1, a0 = mk(0.1), mk(0.5), mk(0.1), mk(0.5)
    ds = torch.from_numpy(g.standard_normal((E, rows)).astype(np.float32) * 0.01).cuda()
    a2 = torch.from_numpy(np.maximum(g.standard_normal((E, rows, 256)), 0).astype(np.float32)).to(torch.float16).cuda()
    stride = 256 * 256 + 1024
    dw2 = torch.full((E, stride), 0.5, device="cuda"); dw1 = torch.full((E, stride), -0.25, device="cuda"); dw3 = torch.full((E, stride), 0.125, device="cuda")
    m2 = w3 = None
    if gen:   # dZ2 = bf16(ds * w3) under the sign words of a2 (what concat_bwd_ws_kernel computes): the reference product uses exactly that
        bits = (a2 > 0)
        wts = (1 << torch.arange(32, device="cuda", dtype=torch.int64))
        m2 = (bits.reshape(E, rows, 8, 32).to(torch.int64) * wts).sum(-1).to(torch.uint32 if hasattr(torch, "uint32") else torch.int64)
        m2 = (bits.reshape(E, rows, 8, 32).to(torch.int64) * wts).sum(-1)
        m2 = torch.where(m2 >= 2 ** 31, m2 - 2 ** 32, m2).to(torch.int32).contiguous()
        w3 = torch.zeros(E, stride, device="cuda")
        w3[:, :256] = torch.from_numpy(g.standard_normal((E, 256)).astype(np.float32)).cuda()
        dz2 = ((ds[:, :, None] * w3[:, None, :256]) * bits).to(torch.bfloat16)
    _lib.check(lib.mimrl_op_concat_dw(stream(), None if gen else P(dz2), P(a1), P(dw2), P(dz1) if two else None, P(a0) if two else None, P(dw1) if two else None,
                                      E, rows, stride, P(ds) if (head or gen) else None, P(a2) if head else None, P(dw3) if head else None,
                                      P(m2) if gen else None, P(w3) if gen else None, None, None, 0))
    torch.cuda.synchronize()
    for name, got, dz, act, base, on in (("dW2", dw2, dz2, a1, 0.5, True), ("dW1", dw1, dz1, a0, -0.25, two)):
        ref = torch.einsum("ekm,ekn->emn", dz.double(), act.double()) + base if on else torch.full((E, 256, 256), base, device="cuda", dtype=torch.float64)
        out = got[:, :65536].reshape(E, 256, 256).double()
        err = (out - ref).abs().max().item()
        scale = ref.abs().max().item()
        assert err <= 3e-6 * scale + 1e-4, f"{name} E={E} rows={rows}: max |err| {err} (scale {scale})"
        assert torch.all(got[:, 65536:] == base), f"{name}: wrote past its 256 x 256 block"
    ref3 = torch.einsum("ek,ekn->en", ds.double(), a2.double()) + 0.125 if head else torch.full((E, 256), 0.125, device="cuda", dtype=torch.float64)
    err = (dw3[:, :256].double() - ref3).abs().max().item()
    assert err <= 3e-6 * ref3.abs().max().item() + 1e-5 * math.sqrt(rows), f"dw3 E={E} rows={rows}: max |err| {err}"
    assert torch.all(dw3[:, 256:] == 0.125), "dw3: wrote past its 256 floats"


@pytest.mark.parametrize("E,B", [(2, 32), (3, 96), (5, 128), (2, 256)])
def test_concat_dw_regenerates_the_separable_first_layer(lib, E, B):
    """concat_dw.hip, a0 regenerated (round 6b): dW1[e] += dZ1[e]^T a0[e] with a0[i B + j] = bf16(relu(P[i] + Q[j])) generated inside the kernel
    from the separable first layer's projections (VMI.py:59-65: the first Linear of f([x_i | y_j]) splits into W0x x_i + (W0y y_j + b0)) --
    the workgroups walk their k-tiles j-block-major with the block's Q rows in LDS; together with the regenerated dZ2 and the score head in
    the same launch (the engine's configuration).  Batch sizes with one, three, four and eight j blocks; reference: float64 product of the
    bf16 a0 the forward kernel would have saved."""
    g = np.random.default_rng(E * 100 + B)
    rows = B * B
    mk = lambda sc: torch.from_numpy(g.standard_normal((E, rows, 256)).astype(np.float32) * sc).to(torch.bfloat16).cuda()
    a1, dz1 = mk(0.5), mk(0.1)
    Pm = torch.from_numpy(g.standard_normal((E, B, 256)).astype(np.float32)).cuda()
    Qm = torch.from_numpy(g.standard_normal((E, B, 256)).astype(np.float32)).cuda()
    a0 = torch.relu(Pm[:, :, None, :] + Qm[:, None, :, :]).to(torch.bfloat16).reshape(E, rows, 256)
    ds = torch.from_numpy(g.standard_normal((E, rows)).astype(np.float32) * 0.01).cuda()
    a2 = torch.from_numpy(np.maximum(g.standard_normal((E, rows, 256)), 0).astype(np.float32)).to(torch.float16).cuda()
    bits = a2 > 0
    wts = 1 << torch.arange(32, device="cuda", dtype=torch.int64)
    m2 = (bits.reshape(E, rows, 8, 32).to(torch.int64) * wts).sum(-1)
    m2 = torch.where(m2 >= 2 ** 31, m2 - 2 ** 32, m2).to(torch.int32).contiguous()
    stride = 256 * 256
    w3 = torch.zeros(E, stride, device="cuda"); w3[:, :256] = torch.from_numpy(g.standard_normal((E, 256)).astype(np.float32)).cuda()
    dz2 = ((ds[:, :, None] * w3[:, None, :256]) * bits).to(torch.bfloat16)
    dw2 = torch.zeros(E, stride, device="cuda"); dw1 = torch.zeros(E, stride, device="cuda"); dw3 = torch.zeros(E, stride, device="cuda")
    _lib.check(lib.mimrl_op_concat_dw(stream(), None, P(a1), P(dw2), P(dz1), None, P(dw1), E, rows, stride, P(ds), P(a2), P(dw3), P(m2), P(w3), P(Pm), P(Qm), B))
    torch.cuda.synchronize()
    for name, got, dz, act in (("dW2", dw2, dz2, a1), ("dW1", dw1, dz1, a0)):
        ref = torch.einsum("ekm,ekn->emn", dz.double(), act.double())
        err = (got.reshape(E, 256, 256).double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        assert err <= 3e-6 * scale + 1e-4, f"{name} E={E} B={B}: max |err| {err} (scale {scale})"
    ref3 = torch.einsum("ek,ekn->en", ds.double(), a2.double())
    assert (dw3[:, :256].double() - ref3).abs().max().item() <= 3e-6 * ref3.abs().max().item() + 1e-5 * B
