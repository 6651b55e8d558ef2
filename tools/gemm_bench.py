"""GPU microbenchmark of libmimrl_hip's strided GEMM on the shapes of the cfg2 step (run on the GPU box)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimrl_amd import _lib
lib = _lib.load(); _lib.check(lib.mimrl_device_check())
S = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def P(t): return C.c_void_p(t.data_ptr())
def run(name, M, N, K, batch, st, a_shape, b_shape, c_shape, prec=1, iters=50, act=0, **kw):
    A = torch.randn(*a_shape, device="cuda"); B = torch.randn(*b_shape, device="cuda"); Cm = torch.zeros(*c_shape, device="cuda")
    arr = (C.c_int64 * 9)(*st)
    f = lambda: lib.mimrl_op_gemm(S, P(A), P(B), P(Cm), M, N, K, batch, arr, None, None, 1.0, 0.0, act, prec)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:34s} {us:8.1f} us  {2.0*M*N*K*batch/us/1e6:8.1f} TF/s")
BT = 6400
run("gx l1: NT 6400x384x256", BT, 384, 256, 1, (256,1,0, 1,256,0, 384,1,0), (BT,256), (384,256), (BT,384))
run("gx l0a: NT 6400x384x74", BT, 384, 74, 1, (74,1,0, 1,74,0, 384,1,0), (BT,74), (384,74), (BT,384))
run("W_t: NT 6400x128x768", BT, 128, 768, 1, (768,1,0, 1,768,0, 128,1,0), (BT,768), (128,768), (BT,128))
run("D-mix: NT 19200x128x128", 19200, 128, 128, 1, (128,1,0, 1,128,0, 128,1,0), (19200,128), (128,128), (19200,128))
run("dh0: NN 6400x256x384", BT, 256, 384, 1, (384,1,0, 256,1,0, 256,1,0), (BT,384), (384,256), (BT,256))
run("L-mix: W[50,50].X_b[50,384] x128", 50, 384, 50, 128, (50,1,0, 384,1,50*384, 384,1,50*384), (50,50), (128,50,384), (128,50,384))
run("tower: NT 128x256x256 x10", 128, 256, 256, 10, (256,1,128*256, 1,256,256*256, 256,1,128*256), (10,128,256), (10,256,256), (10,128,256))
run("cmi l0: NT 128x256x384 x6", 128, 256, 384, 6, (384,1,128*384, 1,384,256*384, 256,1,128*256), (6,128,384), (6,256,384), (6,128,256))
run("wgrad TN 384x256x6400", 384, 256, BT, 1, (1,384,0, 256,1,0, 256,1,0), (BT,384), (BT,256), (384,256), act=256)
run("wgrad TN 128x128x19200", 128, 128, 19200, 1, (1,128,0, 128,1,0, 128,1,0), (19200,128), (19200,128), (128,128), act=256)
run("L-mix wgrad: dY_b.H_b^T x128 ->50x50", 50, 50, 384, 128, (384,1,50*384, 1,384,50*384, 50,1,0), (128,50,384), (128,50,384), (50,50), act=256)

# ---- cfg3-sized (B=256, T=500) and concat-critic shapes
BT3 = 128000
run("cfg3 gx l1: NT 128000x384x256 x4", BT3, 384, 256, 4, (256,1,0, 1,256,384*256, 384,1,BT3*384), (BT3,256), (4,384,256), (4,BT3,384), iters=10)
run("cfg3 W_t: NT 128000x128x768", BT3, 128, 768, 1, (768,1,0, 1,768,0, 128,1,0), (BT3,768), (128,768), (BT3,128), iters=10)
run("cfg3 dh0: NN 128000x256x384 x2", BT3, 256, 384, 2, (384,1,BT3*384, 256,1,384*256, 256,1,BT3*256), (2,BT3,384), (2,384,256), (2,BT3,256), iters=10)
run("cfg3 wgrad TN 384x256x128000", 384, 256, BT3, 1, (1,384,0, 256,1,0, 256,1,0), (BT3,384), (BT3,256), (384,256), act=256, iters=10)
run("concat tail: NT 65536x256x256 x5", 65536, 256, 256, 5, (256,1,65536*256, 1,256,256*256, 256,1,65536*256), (5,65536,256), (5,256,256), (5,65536,256), iters=10)
run("concat dgrad: NN 65536x256x256 x5", 65536, 256, 256, 5, (256,1,65536*256, 256,1,256*256, 256,1,65536*256), (5,65536,256), (5,256,256), (5,65536,256), iters=10)
run("concat wgrad: TN 256x256x65536 x5", 256, 256, 65536, 5, (1,256,65536*256, 256,1,65536*256, 256,1,256*256), (5,65536,256), (5,65536,256), (5,256,256), act=256, iters=10)
run("cfg3 L-mix: W[50,500].X_b[500,384] x256", 50, 384, 500, 256, (500,1,0, 384,1,500*384, 384,1,50*384), (50,500), (256,500,384), (256,50,384), iters=10)
run("cfg3 D-axis wgrad TN 128x128x38400", 128, 128, 38400, 1, (1,128,0, 128,1,0, 128,1,0), (38400,128), (38400,128), (128,128), act=256, iters=10)
run("cfg3 concat dW0 TN 256x128x256 x5", 256, 128, 256, 5, (1,256,256*256, 128,1,2*256*128, 256,1,256*256), (5,256,256), (5,2,256,128), (5,256,256), act=0, iters=10)


# ---- round 5: the same cfg3 products with BOTH operands stored in 16 bits, k-contiguous (csrc/gemm_tall.hip when M >= 16384)
def run16(name, M, N, K, batch, st, a_shape, w_shape, c_shape, flags, batch_in=0, st_bo=None, K2=0, st2=None, a2=None, w2=None, iters=10, bias=False):
    t16 = torch.float16 if flags & 4 else torch.bfloat16
    A = torch.randn(*a_shape, device="cuda").to(t16); W = (torch.randn(*w_shape, device="cuda") * 0.1).to(t16)
    Cm = torch.zeros(*c_shape, device="cuda", dtype=torch.float16 if flags & 8 else torch.float32)
    bv = torch.randn(batch * N, device="cuda") if bias else None
    arr = (C.c_int64 * 9)(*st); arr2 = (C.c_int64 * 6)(*st2) if st2 else None; arrb = (C.c_int64 * 5)(*st_bo) if st_bo else None
    A2 = a2(A) if a2 else None; W2 = w2(W) if w2 else None
    f = lambda: lib.mimrl_op_gemm16(S, P(A), P(W), P(Cm), M, N, K, batch, arr, P(A2) if A2 is not None else None, P(W2) if W2 is not None else None, K2, arr2,
                                    batch_in, arrb, P(bv) if bv is not None else None, flags)
    for _ in range(3): _lib.check(f())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    flops = 2.0 * M * N * (K + K2) * batch
    byts = A.numel() * 2 * (M if flags & 16 else K) / A.shape[-1] + W.numel() * 2 + Cm.numel() * Cm.element_size()
    print(f"{name:58s} {us:8.1f} us  {flops/us/1e6:8.1f} TF/s  {byts/us/1e6:6.2f} TB/s")

for tall in ("1", "0"):
    if tall == "0":
        if os.environ.get("MIMRL_NO_GEMM_TALL"): break
        print("(re-run with MIMRL_NO_GEMM_TALL=1 for the round-4 kernels on the same descriptors)"); break
    run16("cfg3 gx l1 f16s: 128000x384x256 x(2 mod x 2 dir), fp32 out", BT3, 384, 256, 4, (256,1,0, 1,256,384*256, 384,1,BT3*384), (2,BT3,256), (2,2,384,256), (2,2,BT3,384), 7,
          batch_in=2, st_bo=(BT3*256, 2*384*256, 2*BT3*384, 2*384, 384), bias=True)
    run16("cfg3 gx l1 f16s: same, fp16 out", BT3, 384, 256, 4, (256,1,0, 1,256,384*256, 384,1,BT3*384), (2,BT3,256), (2,2,384,256), (2,2,BT3,384), 15,
          batch_in=2, st_bo=(BT3*256, 2*384*256, 2*BT3*384, 2*384, 384), bias=True)
    run16("cfg3 dh0 bf16 (KC,KC): 128000x256x(384+384) x2 mod", BT3, 256, 384, 2, (512,1,2*BT3*512, 1,768,256*768, 256,1,BT3*256), (2,2,BT3,512), (2,256,768), (2,BT3,256), 3,
          K2=384, st2=(512,1,2*BT3*512, 1,768,256*768), a2=lambda A: A[:, 1], w2=lambda W: W[:, :, 384:])
    run16("cfg3 dW_ih l1 TN bf16: 384x256x128000 x(2x2), shared B", 384, 256, BT3, 4, (1,512,BT3*512, 256,1,0, 256,1,384*256), (2,2,BT3,512), (2,BT3,256), (2,2,384,256), 3 | 16,
          batch_in=2, st_bo=(2*BT3*512, BT3*256, 2*384*256, 0, 0))
    run16("cfg3 dW_hh l1 TN bf16 (gap): 384x128x128000 x(2x2)", 384, 128, BT3, 4, (1,512,BT3*512, 128,1,BT3*128, 128,1,384*128), (2,2,BT3,512), (2,2,BT3,128), (2,2,384,128), 3 | 16 | (256 << 8) | (128 << 20),
          batch_in=2, st_bo=(2*BT3*512, 2*BT3*128, 2*384*128, 0, 0))
    run16("cfg2 gx l1 f16s: 6400x384x256 x4 (below the tall threshold)", BT, 384, 256, 4, (256,1,0, 1,256,384*256, 384,1,BT*384), (2,BT,256), (2,2,384,256), (2,2,BT,384), 7,
          batch_in=2, st_bo=(BT*256, 2*384*256, 2*BT*384, 2*384, 384), bias=True, iters=50)
