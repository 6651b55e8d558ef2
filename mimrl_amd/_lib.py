"""ctypes binding of libmimrl_hip.so (include/mimrl.h).  There is NO fallback: if the HIP library is missing or
no gfx950 device is visible, every compute entry point raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIMRL_LIB_PATH: another build of the library -- in practice the host-only AddressSanitizer build (`make -C mimrl_amd/csrc asan`,
# tools/asan_host.sh), which carries the layout / error / KDTree entry points only: the bindings of symbols it lacks are skipped
# MIMRL_DETERMINISTIC=1: the deterministic build (`make -C mimrl_amd/csrc det`; csrc/det.h): same sources, every float atomic replaced by
# order-independent 64-bit fixed-point accumulation, one stream.  Missing library = error, as for the default one.
DETERMINISTIC = os.environ.get("MIMRL_DETERMINISTIC", "0") not in ("", "0")
LIB_PATH = os.environ.get("MIMRL_LIB_PATH") or os.path.join(_HERE, "libmimrl_hip_det.so" if DETERMINISTIC else "libmimrl_hip.so")
MAX_BLOCKS = 4
NSCALARS = 64
PHASES = ["gemm_misc", "gru_fwd", "gru_bwd", "cube_fwd", "cube_bwd", "est_fwd", "est_bwd", "opt", "model_misc"]
S1_LOSS, S1_MIS, S1_LOSSES, S2_LOSS, S2_TASK, S2_MIS, S2_LOSSES = 0, 1, 12, 32, 33, 34, 42

BOUNDS = {"infonce": 0, "nwj": 1, "tuba": 2, "dv": 3, "js_fgan": 4, "js": 5, "smile": 6, "mine": 7, "interpolate": 8}
ACTS = {"none": 0, "relu": 1, "gelu": 2, "tanh": 3}
# (the two GRU bits go together: the forward kernel writes the gate slab in the format the BPTT kernel of the same mode reads)
PREC = {"fp32": 0, "bf16": 15, "bf16_gemm_fwd": 1, "bf16_gemm_bwd": 2, "bf16_gemm": 3, "bf16_gru": 12, "bf16_nogemmbwd": 13}


class MimrlError(RuntimeError):
    pass


class Cfg(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("seq_len", C.c_int32), ("time_len", C.c_int32),
        ("d_t", C.c_int32), ("d_a", C.c_int32), ("d_v", C.c_int32), ("d_common", C.c_int32),
        ("n_blocks", C.c_int32),
        ("d_hiddens", (C.c_int32 * 3) * MAX_BLOCKS), ("d_outs", (C.c_int32 * 3) * MAX_BLOCKS),
        ("res_project", C.c_int32 * MAX_BLOCKS),
        ("bias", C.c_int32), ("ln_first", C.c_int32), ("activation", C.c_int32),
        ("compose_t_sum", C.c_int32), ("compose_k_sum", C.c_int32),
        ("critic_type", C.c_int32), ("bound_type", C.c_int32), ("cmi_hardtanh", C.c_int32),
        ("k_neighbor", C.c_int32), ("bank_capacity", C.c_int32),
        ("dropout", C.c_float * 4), ("dropout_mlp", C.c_float * 3),
        ("coef1", C.c_float * 11), ("coef2", C.c_float * 8),
        ("weight_decay", C.c_float), ("grad_clip", C.c_float),
        ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("precision", C.c_int32), ("use_graph", C.c_int32), ("device_anchors", C.c_int32), ("baseline_type", C.c_int32),
        ("encoder", C.c_int32),
        ("seed", C.c_uint64),
    ]


_FP = C.c_void_p


class Buffers(C.Structure):
    _fields_ = [(n, _FP) for n in (
        "main_p", "main_g", "main_m", "main_v", "crit_p", "crit_g", "crit_m", "crit_v",
        "text", "audio", "video", "labels", "bank_c", "bank_f", "bank_t", "bank_a", "bank_v",
        "anchors", "lr_main", "lr_critic", "pred", "feats", "scalars", "knn_override", "counters")]


_lib = None


def load() -> C.CDLL:
    """Load the HIP library (works on a GPU-less host too: only the layout functions are callable there)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MimrlError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         f"or `make -C mimrl_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch first: it ships its own libamdhip64, and whichever HIP runtime is loaded FIRST is the one libmimrl_hip.so binds to.  Loaded before
    # torch, the library bound to /opt/rocm's runtime while torch used its bundled one -- two runtimes in one process, and mimrl_create failed with
    # "no HIP device visible" (found by running __graft_entry__.build() and smoke() in one process).  The engine holds its tensors in torch anyway.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    if os.environ.get("MIMRL_LIB_PATH"):
        class _Partial:                      # attribute access on a missing symbol yields a throw-away object for the argtypes below
            def __init__(self, lib):
                object.__setattr__(self, "_lib", lib)

            def __getattr__(self, name):
                try:
                    return getattr(self._lib, name)
                except AttributeError:
                    return type("_Missing", (), {})()
        real, lib = lib, _Partial(lib)
    lib.mimrl_last_error.restype = C.c_char_p
    lib.mimrl_bucket_floats.restype = C.c_int64
    lib.mimrl_workspace_bytes.restype = C.c_int64
    lib.mimrl_op_gru_saved_floats.restype = C.c_int64
    lib.mimrl_destroy.restype = None
    lib.mimrl_op_gemm.argtypes = [_FP, _FP, _FP, _FP, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), _FP, _FP,
                                  C.c_float, C.c_float, C.c_int, C.c_int]
    lib.mimrl_op_gemm_ex.argtypes = [_FP, _FP, _FP, _FP, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), _FP, _FP, C.c_int,
                                     C.POINTER(C.c_int64), C.c_int, C.c_int, _FP, _FP, _FP, C.c_int, C.c_int]
    lib.mimrl_comm_unique_id.argtypes = [_FP]
    lib.mimrl_set_comm.argtypes = [_FP, _FP, C.c_int, C.c_int]
    lib.mimrl_set_comm_critic_bf16.argtypes = [_FP, C.c_int]
    lib.mimrl_main_late_offset.argtypes = [_FP]
    lib.mimrl_main_late_offset.restype = C.c_int64
    lib.mimrl_op_gemm16.argtypes = [_FP, _FP, _FP, _FP, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), _FP, _FP, C.c_int,
                                    C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int64), _FP, C.c_int]
    lib.mimrl_op_gemm_wgrad_group.argtypes = [_FP, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_int]
    lib.mimrl_op_adam.argtypes = [_FP, _FP, _FP, _FP, _FP, C.c_int64, _FP, _FP, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_float]
    lib.mimrl_op_gru_forward.argtypes = [_FP] * 11 + [C.c_int, C.c_int, C.c_int]
    lib.mimrl_op_gru_backward.argtypes = [_FP] * 12 + [C.c_int, C.c_int, C.c_int]
    lib.mimrl_op_concat_dw.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int64, C.c_int64] + [C.c_void_p] * 7 + [C.c_int]
    lib.mimrl_op_gru_wgrad.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 5 + [C.c_int64, C.c_int]
    lib.mimrl_op_mi_bound.argtypes = [_FP] * 5 + [C.c_int, C.c_int, C.c_int]
    lib.mimrl_op_mi_bound_ex.argtypes = [_FP] * 6 + [C.c_int, C.c_int, C.c_int, C.c_uint32]
    lib.mimrl_op_mi_sep_infonce.argtypes = [_FP] * 6 + [C.c_int, C.c_int, C.c_int]
    lib.mimrl_op_mi_bound_baseline.argtypes = [_FP] * 7 + [C.c_int, C.c_int, C.c_int]
    lib.mimrl_op_knn.argtypes = [_FP, _FP, C.c_int, C.c_int, _FP, C.c_int, C.c_int, _FP]
    lib.mimrl_op_cmi_loss.argtypes = [_FP] * 7 + [C.c_int, C.c_int, C.c_int]
    lib.mimrl_op_sample_anchors.argtypes = [_FP, _FP, C.c_int, C.c_int, C.c_int, C.c_uint64, _FP, C.c_uint32, C.c_int]
    lib.mimrl_create.argtypes = [C.POINTER(Cfg), _FP, C.POINTER(_FP)]
    lib.mimrl_bind.argtypes = [_FP, C.POINTER(Buffers)]
    lib.mimrl_profile_read.argtypes = [_FP, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    lib.mimrl_profile_read_gemm.argtypes = [_FP, C.POINTER(C.c_double)]
    for fn in ("mimrl_set_bank_rows", "mimrl_stage_grads", "mimrl_stage_grads_part", "mimrl_stage_apply", "mimrl_estimate", "mimrl_profile_enable", "mimrl_set_stage2_prefetch"):
        getattr(lib, fn).argtypes = [_FP, C.c_int]
    lib.mimrl_set_grad_scale.argtypes = [_FP, C.c_float]
    lib.mimrl_stage_grads_part.argtypes = [_FP, C.c_int, C.c_int]
    lib.mimrl_set_inputs.argtypes = [_FP, C.c_int, _FP, _FP, _FP, _FP]
    lib.mimrl_knn_r1_host.argtypes = [_FP, C.c_int, _FP, C.c_int, C.c_int, _FP]
    lib.mimrl_set_knn_override_mask.argtypes = [_FP, C.c_int, C.c_uint]
    for fn in ("mimrl_stage1_step", "mimrl_stage2_step", "mimrl_two_stage_step", "mimrl_destroy", "mimrl_workspace_bytes", "mimrl_params_changed",
               "mimrl_stage2_forward_tail"):
        getattr(lib, fn).argtypes = [_FP]
    lib.mimrl_forward.argtypes = [_FP, C.c_int, C.c_int]
    lib.mimrl_set_kernel_stamps.argtypes = [_FP, _FP, C.c_int]
    lib.mimrl_probe_cube.argtypes = [_FP] * 5
    lib.mimrl_probe_mi.argtypes = [_FP, C.c_int, _FP, _FP, _FP]
    lib.mimrl_probe_cmi.argtypes = [_FP, C.c_int, _FP, _FP, _FP, _FP]
    lib.mimrl_probe_knn.argtypes = [_FP, C.c_int, _FP]
    lib.mimrl_probe_encoders.argtypes = [_FP] * 4
    lib.mimrl_layout_count.argtypes = [C.POINTER(Cfg)]
    lib.mimrl_layout_entry_dim2.argtypes = [C.POINTER(Cfg), C.c_int]
    lib.mimrl_bucket_floats.argtypes = [C.POINTER(Cfg), C.c_int]
    lib.mimrl_layout_entry.argtypes = [C.POINTER(Cfg), C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    if os.environ.get("MIMRL_LIB_PATH"):
        lib = real
    _lib = lib
    return lib


ENCODERS = {"gru": 0, "conv": 1, "lstm": 2}
BASELINES = {"constant": 0, "gaussain": 1, "unnormalized": 2}                 # MIMRL_BASELINE_* (the reference spells it "gaussain")                                             # MIMRL_ENCODER_*

EXPORTS = [
    "mimrl_last_error", "mimrl_abi_version", "mimrl_deterministic", "mimrl_device_check", "mimrl_layout_count", "mimrl_layout_entry", "mimrl_layout_entry_dim2",
    "mimrl_bucket_floats", "mimrl_create", "mimrl_bind", "mimrl_set_bank_rows", "mimrl_set_inputs", "mimrl_stage1_step", "mimrl_stage2_step", "mimrl_two_stage_step",
    "mimrl_stage_grads", "mimrl_stage_grads_part", "mimrl_stage_apply", "mimrl_forward", "mimrl_estimate", "mimrl_profile_enable", "mimrl_profile_read", "mimrl_profile_read_gemm",
    "mimrl_workspace_bytes", "mimrl_params_changed", "mimrl_set_stage2_prefetch", "mimrl_stage2_forward_tail", "mimrl_set_grad_scale", "mimrl_destroy", "mimrl_op_gemm", "mimrl_op_gemm_ex", "mimrl_op_gemm16", "mimrl_op_gemm_wgrad_group",
    "mimrl_op_gru_saved_floats", "mimrl_op_gru_forward", "mimrl_op_gru_backward", "mimrl_op_gru_wgrad", "mimrl_op_concat_dw", "mimrl_op_mi_bound", "mimrl_op_mi_bound_ex", "mimrl_op_mi_bound_baseline", "mimrl_op_mi_sep_infonce", "mimrl_op_knn",
    "mimrl_op_cmi_loss", "mimrl_op_sample_anchors", "mimrl_knn_r1_host", "mimrl_set_knn_override_mask", "mimrl_op_mlp_stack_forward", "mimrl_op_mlp_stack_backward", "mimrl_op_adam",
    "mimrl_probe_cube", "mimrl_probe_mi", "mimrl_probe_cmi", "mimrl_probe_knn", "mimrl_probe_encoders", "mimrl_set_kernel_stamps", "mimrl_comm_unique_id", "mimrl_set_comm", "mimrl_main_late_offset", "mimrl_set_comm_critic_bf16",
    "mimrl_stage1_pipe_prime", "mimrl_stage1_pipe",
]


def check(rc: int) -> int:
    if rc < 0:
        raise MimrlError(f"libmimrl_hip error {rc}: {load().mimrl_last_error().decode()}")
    return rc


def make_cfg(opt, d_t: int, d_a: int, d_v: int, seq_len: int = None, bank_capacity: int = 0, precision: str = "fp32",
             use_graph: bool = False, seed: int = 0, device_anchors: bool = False, batch: int = None) -> Cfg:
    """Translate the reference's ``opt`` Namespace (Parameters.py) into the C config."""
    c = Cfg()
    c.batch = int(batch if batch is not None else opt.batch_size)
    c.time_len = int(opt.time_len)
    c.seq_len = int(seq_len if seq_len is not None else opt.time_len)
    c.d_t, c.d_a, c.d_v, c.d_common = int(d_t), int(d_a), int(d_v), int(opt.d_common)
    enc = getattr(opt, "encoders", "gru")
    if enc not in ENCODERS:
        raise MimrlError(f"--encoders {enc}: choose gru, conv or lstm (Model.py:237)")
    c.encoder = ENCODERS[enc]
    nb = len(opt.d_hiddens)
    if nb > MAX_BLOCKS or len(opt.d_outs) != nb or len(opt.res_project) != nb:
        raise MimrlError("d_hiddens / d_outs / res_project must have the same length (<= 4)")   # MLPProcess.py:129
    c.n_blocks = nb
    for i in range(nb):
        for ax in range(3):
            c.d_hiddens[i][ax] = int(opt.d_hiddens[i][ax])
            c.d_outs[i][ax] = int(opt.d_outs[i][ax])
        c.res_project[i] = int(bool(opt.res_project[i]))
    c.bias, c.ln_first = int(bool(opt.bias)), int(bool(opt.ln_first))
    if opt.activate not in ACTS:
        raise MimrlError(f"--activate {opt.activate} not supported on the HIP path (gelu/relu/tanh)")
    c.activation = ACTS[opt.activate]
    for name, attr in (("features_compose_t", "compose_t_sum"), ("features_compose_k", "compose_k_sum")):
        v = getattr(opt, name, "mean")
        if v not in ("mean", "sum"):
            raise MimrlError(f"--{name} {v}: only mean/sum supported")
        setattr(c, attr, int(v == "sum"))
    if opt.critic_type not in ("separate", "concat"):
        raise NotImplementedError(opt.critic_type)             # VMI.py:44-45
    c.critic_type = 0 if opt.critic_type == "separate" else 1
    bl = getattr(opt, "baseline_type", "constant")
    if bl not in BASELINES:
        raise NotImplementedError(f"--baseline_type {bl}")                           # VMI.py:89-90
    c.baseline_type = BASELINES[bl]
    if opt.bound_type not in BOUNDS:
        raise NotImplementedError(opt.bound_type)              # Model.py:144-145
    c.bound_type = BOUNDS[opt.bound_type]
    if opt.cmi_last_acticate not in ("sigmoid", "hardtanh"):
        raise NotImplementedError(opt.cmi_last_acticate)       # Model.py:62-63
    c.cmi_hardtanh = int(opt.cmi_last_acticate == "hardtanh")
    c.k_neighbor = int(opt.k_neighbor)
    c.bank_capacity = int(bank_capacity)
    for i in range(4):
        c.dropout[i] = float(opt.dropout[i])
    for i in range(3):
        c.dropout_mlp[i] = float(opt.dropout_mlp[i])
    if len(opt.loss_mi_coefficient1) != 11 or len(opt.loss_mi_coefficient2) != 8:
        raise MimrlError("loss_mi_coefficient1/2 need 11 / 8 entries")
    for i in range(11):
        c.coef1[i] = float(opt.loss_mi_coefficient1[i])
    for i in range(8):
        c.coef2[i] = float(opt.loss_mi_coefficient2[i])
    c.weight_decay, c.grad_clip = float(opt.weight_decay), float(opt.gradient_clip)
    c.beta1, c.beta2, c.adam_eps = 0.9, 0.999, 1e-8
    c.precision = PREC[precision]
    c.use_graph = int(bool(use_graph))
    c.device_anchors = int(bool(device_anchors))
    c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return c


def layout_entries(cfg: Cfg) -> Tuple[List[Tuple[str, int, int, Tuple[int, ...]]], Tuple[int, int]]:
    """-> ([(name, group, offset, shape)], (main_floats, critic_floats)) as the native library lays them out."""
    lib = load()
    n = check(lib.mimrl_layout_count(C.byref(cfg)))
    out = []
    name = C.create_string_buffer(256)
    g, nd, d0, d1 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    off = C.c_int64()
    for i in range(n):
        check(lib.mimrl_layout_entry(C.byref(cfg), i, name, 256, C.byref(g), C.byref(off), C.byref(nd), C.byref(d0), C.byref(d1)))
        shape = (d0.value, d1.value) if nd.value == 2 else (d0.value,)
        if nd.value == 3:
            shape = (d0.value, d1.value, check(lib.mimrl_layout_entry_dim2(C.byref(cfg), i)))
        out.append((name.value.decode(), g.value, off.value, shape))
    sizes = (check(lib.mimrl_bucket_floats(C.byref(cfg), 0)), check(lib.mimrl_bucket_floats(C.byref(cfg), 1)))
    return out, sizes


def knn_r1_host(z, anchors, k: int):
    """k nearest non-anchor rows of the 1-column bank ``z`` for every anchor, scikit-learn KDTree tie order (csrc/knn_r1.cpp).
    Pure host code: callable without a GPU.  -> int32 [m, k] (original bank rows) or None in scikit-learn's brute-force regime."""
    import numpy as np
    z = np.ascontiguousarray(np.asarray(z, dtype=np.float32).reshape(-1))
    a = np.ascontiguousarray(np.asarray(anchors, dtype=np.int32).reshape(-1))
    if int(k) >= (len(z) - len(a)) // 2:
        return None
    out = np.empty((len(a), int(k)), dtype=np.int32)
    check(load().mimrl_knn_r1_host(z.ctypes.data_as(C.c_void_p), len(z), a.ctypes.data_as(C.c_void_p), len(a), int(k),
                                   out.ctypes.data_as(C.c_void_p)))
    return out
