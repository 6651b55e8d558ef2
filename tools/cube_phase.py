"""In-kernel phase times of the fused CubeMLP forward (run on the GPU box with a `make PHASE_PROBE=1` build of the library:
tools/cube_phase.sh).  Workgroup 0 of the last launch of every instantiation leaves 100 MHz ticks at its phase boundaries."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    from mimrl_amd.engine import HipEngine
    from mimrl_amd import synth
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    opt, N = bench.workload(wl)
    B, T = opt.batch_size, opt.time_len
    eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision="bf16", seed=1234, device_anchors=True)
    eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
    eng.set_batch(*synth.synthetic_batch(B, T, seed=0))
    banks = synth.synthetic_banks(N, seed=0)
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    eng.set_stage2_prefetch(1)
    for _ in range(30):
        eng.step()
    torch.cuda.synchronize()
    from mimrl_amd import _lib
    lib = _lib.load()
    buf = (C.c_longlong * 128)()
    lib.mimrl_dbg_cube_phases.argtypes = [C.POINTER(C.c_longlong)]
    assert lib.mimrl_dbg_cube_phases(buf) == 0
    names = ["entry", "tile+L weights", "phase L", "phase K", "Az images", "H = act(Z W1)", "Y = H W2 + Z Wr", "LayerNorm D / end"]
    for save in (0, 1):
        for nmt in (1, 2, 3):
            o = save * 64 + (nmt - 1) * 16
            t = [buf[o + i] for i in range(8)]
            if t[0] == 0:
                continue
            print(f"cube_fwd_fused<save={save}, row tiles={nmt}>: total {(t[7] - t[0]) * 0.01:.2f} us")
            for i in range(1, 8):
                print(f"    {names[i]:22s} {(t[i] - t[i - 1]) * 0.01:7.2f} us")
            y = [buf[o + 13], buf[o + 14]]
            if y[0]:
                print(f"      set-up: entry -> weight / parameter requests issued {(y[0] - t[0]) * 0.01:.2f} | tile requested, arrived, converted, stored {(y[1] - y[0]) * 0.01:.2f} | parameters + weight images committed, barrier {(t[1] - y[1]) * 0.01:.2f} us")
            x = [buf[o + i] for i in range(8, 13)]
            if x[0]:
                print(f"      phase L, first round: transpose {(x[0] - t[1]) * 0.01:.2f} | W1.X + act -> H {(x[1] - x[0]) * 0.01:.2f} | W2.H + Wr.X -> tile {(x[2] - x[1]) * 0.01:.2f} | LayerNorm {(x[3] - x[2]) * 0.01:.2f} us")
                print(f"      H phase: weight images + MFMA {(x[4] - t[4]) * 0.01:.2f} | act epilogue {(t[5] - x[4]) * 0.01:.2f} us")

    b = (C.c_longlong * 64)()
    lib.mimrl_dbg_cube_bwd_phases.argtypes = [C.POINTER(C.c_longlong)]
    assert lib.mimrl_dbg_cube_bwd_phases(b) == 0
    ln = ["entry", "u requests + zero fill", "weight scatter", "LN pass 1", "LN pass 2 -> dY", "dU = W2^T dY * act'", "dX = W1^T dU + Wr^T dY", "bias atomics / end"]
    for base, what in ((0, "laxis_bwd (ol <= 32: block 2)"), (16, "laxis_bwd (ol > 32: block 1)")):
        tt = [b[base + i] for i in range(8)]
        if tt[0]:
            print(f"{what}: total {(tt[7] - tt[0]) * 0.01:.2f} us")
            for i in range(1, 8):
                print(f"    {ln[i]:26s} {(tt[i] - tt[i - 1]) * 0.01:7.2f} us")
    dn = ["entry", "requests (24 fragments, u)", "LayerNorm D backward", "dU = dY W2 * act'", "dX = dU W1 + dY Wr / end"]
    for base, what in ((32, "daxis_bwd (<= 200 workgroups: block 2)"), (48, "daxis_bwd (block 1)")):
        tt = [b[base + i] for i in range(5)]
        if tt[0]:
            print(f"{what}: total {(tt[4] - tt[0]) * 0.01:.2f} us")
            for i in range(1, 5):
                print(f"    {dn[i]:26s} {(tt[i] - tt[i - 1]) * 0.01:7.2f} us")
    k = (C.c_longlong * 16)()
    lib.mimrl_dbg_kmix_phases.argtypes = [C.POINTER(C.c_longlong)]
    assert lib.mimrl_dbg_kmix_phases(k) == 0
    kn = ["entry", "weights -> LDS, zero accumulators", "element loop", "wave reductions + LDS atomics", "barrier", "global atomics / end"]
    for base, what in ((0, "kmix_bwd (block 2)"), (8, "kmix_bwd (block 1)")):
        tt = [k[base + i] for i in range(6)]
        if tt[0]:
            print(f"{what}: total {(tt[5] - tt[0]) * 0.01:.2f} us")
            for i in range(1, 6):
                print(f"    {kn[i]:34s} {(tt[i] - tt[i - 1]) * 0.01:7.2f} us")
    q = (C.c_longlong * 8)()
    lib.mimrl_dbg_nce_phases.argtypes = [C.POINTER(C.c_longlong)]
    assert lib.mimrl_dbg_nce_phases(q) == 0
    nn = ["entry", "stage g(x), h(y) tile", "scores (MFMA) -> LDS", "row log-sum-exps", "dS in place", "dh tile = dS . g", "dg += dS^T . h (atomics) / end"]
    print(f"mi_sep_nce: total {(q[6] - q[0]) * 0.01:.2f} us")
    for i in range(1, 7):
        print(f"    {nn[i]:34s} {(q[i] - q[i - 1]) * 0.01:7.2f} us")
    mo = (C.c_longlong * 16)()
    lib.mimrl_dbg_model_ops_phases.argtypes = [C.POINTER(C.c_longlong)]
    assert lib.mimrl_dbg_model_ops_phases(mo) == 0
    print(f"tail_pre (sample 0, audio slot): step counter read {(mo[1] - mo[0]) * 0.01:.2f} | rows {(mo[2] - mo[1]) * 0.01:.2f} | means {(mo[3] - mo[2]) * 0.01:.2f} us")
    print(f"ln_relu_drop_bwd16 (workgroup 0): zero + barrier {(mo[9] - mo[8]) * 0.01:.2f} | rows {(mo[10] - mo[9]) * 0.01:.2f} | parameter-gradient shuffles + LDS atomics + barrier {(mo[11] - mo[10]) * 0.01:.2f} | global atomics {(mo[12] - mo[11]) * 0.01:.2f} us")


if __name__ == "__main__":
    main()
