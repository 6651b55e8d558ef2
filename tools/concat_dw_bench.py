"""Stand-alone timing of the concat critic's hidden-layer weight gradients: the one-launch kernel (concat_dw.hip) against the two split-K
GEMMs it replaces.  GPU box: python tools/concat_dw_bench.py [B] [E]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mimrl_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
E = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = B * B
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda sc: (torch.randn(E, rows, 256, device="cuda", generator=g) * sc).to(torch.bfloat16)
dz2, a1, dz1, a0 = mk(0.1), mk(0.5), mk(0.1), mk(0.5)
dw2 = torch.zeros(E, 65536, device="cuda"); dw1 = torch.zeros(E, 65536, device="cuda"); dw3 = torch.zeros(E, 65536, device="cuda")
ds = torch.randn(E, rows, device="cuda", generator=g) * 0.01
a2 = torch.relu(torch.randn(E, rows, 256, device="cuda", generator=g)).to(torch.float16)
S = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def one():
    _lib.check(lib.mimrl_op_concat_dw(S, P(dz2), P(a1), P(dw2), P(dz1), P(a0), P(dw1), E, rows, 65536, None, None, None, None, None, None, None, 0))

def one3():
    _lib.check(lib.mimrl_op_concat_dw(S, P(dz2), P(a1), P(dw2), P(dz1), P(a0), P(dw1), E, rows, 65536, P(ds), P(a2), P(dw3), None, None, None, None, 0))

m2 = torch.randint(-2 ** 31, 2 ** 31 - 1, (E, rows, 8), device="cuda", dtype=torch.int32, generator=g)
w3 = torch.randn(E, 65536, device="cuda", generator=g)

def one3g():
    _lib.check(lib.mimrl_op_concat_dw(S, None, P(a1), P(dw2), P(dz1), P(a0), P(dw1), E, rows, 65536, P(ds), P(a2), P(dw3), P(m2), P(w3), None, None, 0))

Pm = torch.randn(E, B, 256, device="cuda", generator=g); Qm = torch.randn(E, B, 256, device="cuda", generator=g)

def one3ga():
    _lib.check(lib.mimrl_op_concat_dw(S, None, P(a1), P(dw2), P(dz1), None, P(dw1), E, rows, 65536, P(ds), P(a2), P(dw3), P(m2), P(w3), P(Pm), P(Qm), B))

def gemm(A, Bm, Cm):
    K = rows
    _lib.check(lib.mimrl_op_gemm16(S, P(A), P(Bm), P(Cm), 256, 256, K, E, (C.c_int64 * 9)(1, 256, K * 256, 256, 1, K * 256, 256, 1, 65536), None, None, 0, None, 0, None, None, 3 | 16))

def pair():
    gemm(dz2, a1, dw2); gemm(dz1, a0, dw1)

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

one(); torch.cuda.synchronize(); r = (dw2.clone(), dw1.clone()); dw2.zero_(); dw1.zero_()
pair(); torch.cuda.synchronize()
print("max |one launch - GEMM pair|: dW2 %.3e (scale %.3e)  dW1 %.3e (scale %.3e)" % ((r[0] - dw2).abs().max().item(), dw2.abs().max().item(), (r[1] - dw1).abs().max().item(), dw1.abs().max().item()))
by = 4 * E * rows * 256 * 2
for name, fn in (("one launch (concat_dw)", one), ("two split-K GEMMs", pair), ("one launch (concat_dw)", one), ("one launch + score head", one3), ("  ... + dZ2 regenerated", one3g), ("  ... + a0 regenerated", one3ga)):
    us = timed(fn)
    print("%-24s %8.1f us   %.2f TB/s of the %.0f MB of operands" % (name, us, by / us / 1e6, by / 1e6))
