"""What does the vendor GEMM library reach on the big cfg3 products?  (torch.matmul -> hipBLASLt / rocBLAS; GPU box)"""
import torch

def t(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = "cuda"
for name, M, N, K, dt in (("dh0 (bf16 x bf16)", 128000, 256, 768, torch.bfloat16), ("l1 projection (fp16 x fp16)", 128000, 768, 256, torch.float16),
                          ("l0 projection", 128000, 768, 80, torch.float16), ("cfg2 l1 projection", 6400, 768, 256, torch.float16),
                          ("cfg2 dh0", 6400, 256, 768, torch.bfloat16)):
    A = torch.randn(2, M, K, device=dev, dtype=dt)
    B = torch.randn(2, K, N, device=dev, dtype=dt)
    us = t(lambda: torch.bmm(A, B))
    fl = 2.0 * 2 * M * N * K
    print("%-30s M %6d N %4d K %4d batch 2: %8.1f us  %6.1f TFLOP/s  (16-bit output)" % (name, M, N, K, us, fl / us / 1e6))
    C = torch.empty(2, M, N, device=dev, dtype=torch.float32)
    try:
        us = t(lambda: torch.bmm(A.float(), B.float(), out=C))
        print("%-30s fp32 in/out (incl. conversions): %8.1f us" % ("", us))
    except Exception as e:
        print("fp32:", e)
# weight gradient: [384 x 128000] x [128000 x 256]
A = torch.randn(4, 384, 128000, device=dev, dtype=torch.bfloat16); B = torch.randn(4, 128000, 256, device=dev, dtype=torch.bfloat16)
us = t(lambda: torch.bmm(A, B)); print("wgrad l1 (bf16) 4 x [384 x 128000 x 256]: %8.1f us  %6.1f TFLOP/s" % (us, 2.0 * 4 * 384 * 128000 * 256 / us / 1e6))
