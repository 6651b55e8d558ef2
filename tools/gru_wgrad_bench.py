"""Stand-alone timing of the layer-0 GRU weight gradients: the one-pass kernel (gru_wgrad.hip) against the two batched split-K GEMMs it
replaces (back to back on one stream AND on two streams, as the engine ran them).  GPU box: python tools/gru_wgrad_bench.py [rows] [kp]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mimrl_amd import _lib

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128000
kp = int(sys.argv[2]) if len(sys.argv) > 2 else 80
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
g = torch.Generator(device="cuda").manual_seed(0)
dg = (torch.randn(2, 2, rows, 512, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
x = (torch.randn(2, rows, kp, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
hp = (torch.randn(2, 2, rows, 128, device="cuda", generator=g) * 0.3).to(torch.bfloat16)
dwih = torch.zeros(2, 2, 384, kp, device="cuda"); dwhh = torch.zeros(2, 2, 384, 128, device="cuda")
arr = lambda ts: (C.c_void_p * 4)(*[t.data_ptr() for t in ts])
s0 = torch.cuda.current_stream(); s1 = torch.cuda.Stream()
S = lambda s: C.c_void_p(s.cuda_stream)
seqs = [(m, d) for m in range(2) for d in range(2)]
a_dg, a_x, a_hp = arr([dg[m, d] for m, d in seqs]), arr([x[m] for m, d in seqs]), arr([hp[m, d] for m, d in seqs])
a_ih, a_hh = arr([dwih[m, d] for m, d in seqs]), arr([dwhh[m, d] for m, d in seqs])

def one_pass():
    _lib.check(lib.mimrl_op_gru_wgrad(S(s0), a_dg, a_x, a_hp, a_ih, a_hh, rows, kp))

def gemm(st, A, B, Cm, M, N, K, strides, flags, st_bo):
    _lib.check(lib.mimrl_op_gemm16(S(st), P(A), P(B), P(Cm), M, N, K, 4, (C.c_int64 * 9)(*strides), None, None, 0, None, 2, (C.c_int64 * 5)(*st_bo), None, flags))

def pair(two_streams):
    K = rows
    gemm(s0, dg, x, dwih, 384, kp, K, (1, 512, K * 512, kp, 1, 0, kp, 1, 384 * kp), 3 | 16, (2 * K * 512, K * kp, 2 * 384 * kp, 0, 0))
    st = s1 if two_streams else s0
    if two_streams:
        s1.wait_stream(s0)
    gemm(st, dg, hp, dwhh, 384, 128, K, (1, 512, K * 512, 128, 1, K * 128, 128, 1, 384 * 128), 3 | 16 | (256 << 8) | (128 << 20),
         (2 * K * 512, 2 * K * 128, 2 * 384 * 128, 0, 0))
    if two_streams:
        s0.wait_stream(s1)

def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

one_pass(); torch.cuda.synchronize()
r1 = (dwih.clone(), dwhh.clone()); dwih.zero_(); dwhh.zero_()
pair(False); torch.cuda.synchronize()
print("max |one-pass - GEMM pair|: dW_ih %.3e (scale %.3e)  dW_hh %.3e (scale %.3e)" % ((r1[0] - dwih).abs().max().item(), dwih.abs().max().item(),
                                                                                       (r1[1] - dwhh).abs().max().item(), dwhh.abs().max().item()))
by = 4 * rows * 512 * 2 + 2 * rows * kp * 2 + 4 * rows * 128 * 2
for name, fn in (("one-pass kernel", one_pass), ("GEMM pair, one stream", lambda: pair(False)), ("GEMM pair, two streams", lambda: pair(True)),
                 ("one-pass kernel", one_pass)):
    us = timed(fn)
    print("%-24s %8.1f us   %.2f TB/s of the %.0f MB a single pass reads" % (name, us, by / us / 1e6, by / 1e6))
