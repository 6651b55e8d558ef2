"""Cycle budget of the BPTT cell step from in-kernel stamps (GPU box; needs `make -C mimrl_amd/csrc probe`, loaded through MIMRL_LIB_PATH).
usage: MIMRL_LIB_PATH=mimrl_amd/libmimrl_hip_probe.so python tools/gru_bwd_phase.py [workload] [out.json]
Wave 0 of workgroup (0, 0, 0) of gru_bwd_kernel<bf16, bf16 dg> sums shader-clock differences at seven phase boundaries of every cell step
(gru.hip: GPH); printed per cell step in cycles and in microseconds (the wave's own clock: cycles / (wall time of the launch))."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    from mimrl_amd.engine import HipEngine
    from mimrl_amd import synth, _lib
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    out = sys.argv[2] if len(sys.argv) > 2 else None
    opt, N = bench.workload(wl)
    B, T = opt.batch_size, opt.time_len
    eng = HipEngine(opt, 768, 74, 35, seq_len=T, bank_capacity=N, precision="bf16", use_graph=True, seed=1234, device_anchors=True)
    eng.load_params(synth.default_state([(n, tuple(v.shape)) for n, v in eng.params.items()], 0))
    eng.set_batch(*synth.synthetic_batch(B, T, seed=0))
    banks = synth.synthetic_banks(N, seed=0)
    eng.set_banks(*(banks[k] for k in "CFTAV"))
    eng.set_stage2_prefetch(1)
    for _ in range(20):
        eng.step()
    torch.cuda.synchronize()
    lib = _lib.load()
    buf = (C.c_longlong * 32)()
    lib.mimrl_dbg_gru_bwd_phases.argtypes = [C.POINTER(C.c_longlong)]
    assert lib.mimrl_dbg_gru_bwd_phases(buf) == 0
    names = ["operand wait (vmcnt)", "gate / gradient math + LDS tile writes", "5 global stores issued", "lgkmcnt(0) + s_barrier",
             "3 prefetches issued", "12 fragment reads returned", "24 MFMAs retired"]
    res = {"workload": wl, "T": T, "B": B, "source": "tools/gru_bwd_phase.py: s_memtime differences summed by wave 0 of workgroup (0,0,0) over the cell "
           "steps of the LAST captured step's launches; probe build (make probe): the stamps themselves cost ~10 % of the step"}
    for o, layer in ((0, "layer1"), (16, "layer0")):
        steps = buf[o + 7]
        if not steps:
            continue
        cyc = [buf[o + i] / steps for i in range(7)]
        wall_us = (buf[o + 9] - buf[o + 8]) * 0.01
        tot = sum(cyc)
        ghz = tot * steps / (wall_us * 1e3) if wall_us > 0 else float("nan")
        print(f"gru_bwd_kernel {layer}: {steps} cell steps, {tot:.0f} cycles per step stamped, launch wall {wall_us:.1f} us "
              f"({wall_us / steps:.3f} us per step incl. prologue / epilogue; stamped cycles / wall = {ghz:.2f} GHz)")
        for n_, c in zip(names, cyc):
            print(f"    {n_:42s} {c:7.0f} cycles  {c / tot * 100:5.1f} %   {c / tot * wall_us / steps:6.3f} us")
        res[layer] = {"steps": int(steps), "cycles_per_step": dict(zip(names, [round(c, 1) for c in cyc])), "stamped_cycles_per_step": round(tot, 1),
                      "launch_wall_us": round(wall_us, 2), "us_per_step_wall": round(wall_us / steps, 4)}
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
