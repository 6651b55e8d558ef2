"""Parameter inventory and flat-bucket layout of the hot path.

Names and shapes are exactly the reference's ``state_dict`` entries (Model.py:243-303,
MLPProcess.py:26-52, VMI.py:32-45) so that checkpoints/optimizer splits keep working:  the optimiser split is by
substring -- names containing ``vmi``/``vcmi`` go to the critic bucket, the rest to the main bucket
(Solver.py:124-133).  Every tensor lives at a 64-float-aligned offset inside one of two flat fp32 buffers
("main", "critic"); gradients and both Adam moments use the same offsets in sibling buffers, so one fused
clip+Adam launch and one RCCL all-reduce cover a whole bucket.

The native library re-derives the same table (csrc/layout.cpp); tests/test_layout.py checks they agree.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

ALIGN = 64  # floats (256 B)

VMI_NAMES = ["f_t", "f_a", "f_v", "t_a", "t_v"]
VCMI_NAMES = ["ac_t", "ta_c", "vc_t", "tv_c", "tc_a", "tc_v"]


@dataclass(frozen=True)
class Entry:
    name: str
    shape: Tuple[int, ...]
    group: str      # "main" | "critic"
    offset: int     # in floats

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


def named_shapes(opt, d_t: int, d_a: int, d_v: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (name, shape) list, main tensors first then critics, in reference registration order."""
    D = int(opt.d_common)
    H = D
    out: List[Tuple[str, Tuple[int, ...]]] = []
    enc = getattr(opt, "encoders", "gru")
    if enc == "conv":                                                     # Model.py:247-249 (conv_a before conv_v)
        out += [("conv_a.weight", (D, d_a, 3)), ("conv_a.bias", (D,)), ("conv_v.weight", (D, d_v, 3)), ("conv_v.bias", (D,))]
    elif enc not in ("gru", "lstm"):
        raise NotImplementedError(f"--encoders {enc}")
    # rnn_v is registered before rnn_a in the reference (Model.py:251-252 lstm: 1 layer, 4 gates; :254-255 gru: 2 layers, 3 gates)
    ng, nlayers = (3, 2) if enc == "gru" else (4, 1) if enc == "lstm" else (0, 0)
    for mod, d in (("rnn_v", d_v), ("rnn_a", d_a)):
        for layer in range(nlayers):
            din = d if layer == 0 else 2 * H
            for sfx in ("", "_reverse"):
                out += [(f"{mod}.weight_ih_l{layer}{sfx}", (ng * H, din)), (f"{mod}.weight_hh_l{layer}{sfx}", (ng * H, H)),
                        (f"{mod}.bias_ih_l{layer}{sfx}", (ng * H,)), (f"{mod}.bias_hh_l{layer}{sfx}", (ng * H,))]
    out += [("ln_a.weight", (D,)), ("ln_a.bias", (D,)), ("ln_v.weight", (D,)), ("ln_v.bias", (D,))]
    out += [("W_t.weight", (D, d_t))]
    d_in = [int(opt.time_len), 3, D]
    for i, (hid, dout) in enumerate(zip(opt.d_hiddens, opt.d_outs)):
        pre = f"mlp_encoder.layers_stack.{i}"
        for ax, nm in enumerate("lkd"):
            out.append((f"{pre}.mlp_{nm}.fc1.weight", (hid[ax], d_in[ax])))
            if opt.bias:
                out.append((f"{pre}.mlp_{nm}.fc1.bias", (hid[ax],)))
            out.append((f"{pre}.mlp_{nm}.fc2.weight", (dout[ax], hid[ax])))
            if opt.bias:
                out.append((f"{pre}.mlp_{nm}.fc2.bias", (dout[ax],)))
        for ax, nm in enumerate("lkd"):
            n = d_in[ax] if opt.ln_first else dout[ax]
            out += [(f"{pre}.ln_{nm}.weight", (n,)), (f"{pre}.ln_{nm}.bias", (n,))]
        if opt.res_project[i]:
            for ax, nm in enumerate("lkd"):
                out.append((f"{pre}.res_projection_{nm}.weight", (dout[ax], d_in[ax])))
        d_in = list(dout)
    out += [("classifier.0.weight", (int(getattr(opt, "num_class", 1)), d_in[2])),
            ("classifier.0.bias", (int(getattr(opt, "num_class", 1)),))]
    hid, emb = 256, 128                                                   # Model.py:285
    for n in VMI_NAMES:
        pre = f"vmi_estimator_{n}.critic_model"
        if opt.critic_type == "separate":
            for tw in ("MLP_g", "MLP_h"):
                dims = [(hid, D), (hid, hid), (hid, hid), (emb, hid)]
                for idx, shp in zip((0, 2, 4, 6), dims):
                    out += [(f"{pre}.{tw}.{idx}.weight", shp), (f"{pre}.{tw}.{idx}.bias", (shp[0],))]
        elif opt.critic_type == "concat":
            dims = [(hid, 2 * D), (hid, hid), (hid, hid), (1, hid)]
            for idx, shp in zip((0, 2, 4, 6), dims):
                out += [(f"{pre}.MLP_f.{idx}.weight", shp), (f"{pre}.MLP_f.{idx}.bias", (shp[0],))]
        else:
            raise NotImplementedError(opt.critic_type)                    # VMI.py:44-45
    if getattr(opt, "baseline_type", "constant") == "unnormalized":      # VMI.py:82-84: mlps(128, 256, 1, 2); kept behind all
        for n in VMI_NAMES:                                               # critics so that the towers stay uniformly strided
            pre = f"vmi_estimator_{n}.baseline_model.MLP"
            dims = [(hid, D), (hid, hid), (hid, hid), (1, hid)]
            for idx, shp in zip((0, 2, 4, 6), dims):
                out += [(f"{pre}.{idx}.weight", shp), (f"{pre}.{idx}.bias", (shp[0],))]
    for n in VCMI_NAMES:
        pre = f"vcmi_estimator_{n}.classifier.mlp"
        dims = [(hid, 3 * emb), (hid, hid), (hid, hid), (2, hid)]
        for idx, shp in zip((0, 2, 4, 6), dims):
            out += [(f"{pre}.{idx}.weight", shp), (f"{pre}.{idx}.bias", (shp[0],))]
    return out


def is_critic(name: str) -> bool:
    return ("vmi" in name) or ("vcmi" in name)


def is_late(name: str) -> bool:
    """The layer-0 recurrence tensors: their gradients are final last in the backward pass (behind the layer-0 BPTT), so they sit at the
    TAIL of the main bucket -- everything in front of them is one contiguous range a data-parallel step can all-reduce early."""
    return name.startswith("rnn_") and "_l0" in name


def build_layout(opt, d_t: int, d_a: int, d_v: int) -> Tuple[List[Entry], Dict[str, int]]:
    """-> (entries, sizes) with sizes = {"main": floats, "critic": floats} (each padded to ALIGN).  Entries come in the reference's
    registration order; offsets are assigned in two passes (``is_late`` tensors last), exactly as csrc/layout.cpp does."""
    cursor = {"main": 0, "critic": 0}
    shapes = [(n, tuple(int(s) for s in shp)) for n, shp in named_shapes(opt, d_t, d_a, d_v)]
    offset: Dict[str, int] = {}
    for late in (False, True):
        for name, shape in shapes:
            if is_late(name) != late:
                continue
            g = "critic" if is_critic(name) else "main"
            n = 1
            for s_ in shape:
                n *= s_
            offset[name] = cursor[g]
            cursor[g] += (n + ALIGN - 1) // ALIGN * ALIGN
    entries = [Entry(name, shape, "critic" if is_critic(name) else "main", offset[name]) for name, shape in shapes]
    return entries, dict(cursor)
