"""Where a tile of concat_bwd_kernel spends its time (GPU box; needs `make -C mimrl_amd/csrc probe`, loaded through MIMRL_LIB_PATH).
usage: MIMRL_LIB_PATH=mimrl_amd/libmimrl_hip_probe.so python tools/concat_phase.py [workload] [out.json]
Thread 0 of workgroup 0 sums wall-clock differences (100 MHz) at the phase boundaries of every tile of its run (concat_fused.hip: CPH)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    from mimrl_amd.engine import HipEngine
    from mimrl_amd import synth, _lib
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    out = sys.argv[2] if len(sys.argv) > 2 else None
    opt, N = bench.workload(wl)
    B, T = opt.batch_size, opt.time_len
    eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision="bf16", use_graph=True, seed=1234, device_anchors=True)
    eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
    eng.set_batch(*synth.synthetic_batch(B, T, seed=0))
    banks = synth.synthetic_banks(N, seed=0)
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    eng.set_stage2_prefetch(1)
    lib = _lib.load()
    buf = (C.c_longlong * 16)()
    lib.mimrl_dbg_concat_bwd_phases.argtypes = [C.POINTER(C.c_longlong)]
    for _ in range(6):
        eng.step()
    torch.cuda.synchronize()
    assert lib.mimrl_dbg_concat_bwd_phases(buf) == 0      # read and clear
    nstep = 10
    for _ in range(nstep):
        eng.step()
    torch.cuda.synchronize()
    assert lib.mimrl_dbg_concat_bwd_phases(buf) == 0
    names = ["dZ2 generation (loads -> LDS tile)", "product W2 (8 chunks, 64 MFMAs per wave)", "epilogue 1 (mask, LDS tile, dZ1 out)",
             "product W1", "epilogue 0 (mask, dQ accumulate, dP)", "dQ flush (atomics)"]
    res = {"workload": wl, "B": B, "source": "tools/concat_phase.py, probe build"}
    for o, stage in ((8, "stage 1 (weight-gradient outputs)"), (0, "stage 2")):
        tiles = buf[o + 6]
        if not tiles:
            continue
        us = [buf[o + i] * 0.01 / tiles for i in range(6)]
        print(f"concat_bwd_kernel {stage}: {tiles / nstep:.1f} tiles per launch in workgroup 0, {sum(us):.2f} us per tile")
        for n_, u in zip(names, us):
            print(f"    {n_:46s} {u:7.2f} us  {u / sum(us) * 100:5.1f} %")
        res[stage] = {"tiles_per_launch": tiles / nstep, "us_per_tile": dict(zip(names, [round(u, 3) for u in us]))}
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
