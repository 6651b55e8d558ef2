"""HipEngine: owns the torch tensors (the "tensor-holding boundary") and drives libmimrl_hip through its C ABI.

PyTorch-ROCm is used only for device memory, the stream and (in dist.py) RCCL; every FLOP of the two-stage step
runs in the hand-written HIP library.  There is no eager/CPU fallback anywhere in this module.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import Buffers, MimrlError, check


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class HipEngine:
    """One engine = one (process, GPU).  Mirrors the state the reference keeps in Model + two Adam optimizers."""

    def __init__(self, opt, d_t: int, d_a: int, d_v: int, seq_len: Optional[int] = None, bank_capacity: int = 0,
                 precision: str = "fp32", use_graph: bool = False, seed: int = 0, device: Optional[torch.device] = None,
                 device_anchors: bool = False, share: Optional["HipEngine"] = None, batch: Optional[int] = None):
        """``share``: another engine whose parameter / gradient / Adam buckets, step counters, learning rates and feature
        banks this one binds too (same optimizer, different batch size: the partial last batch of a loader with
        drop_last=False, Parameters.py:21).  ``batch`` overrides ``opt.batch_size``."""
        if not torch.cuda.is_available():
            raise MimrlError("HipEngine needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        self.opt = opt
        if share is not None:
            bank_capacity = share.cfg.bank_capacity
        self.cfg = _lib.make_cfg(opt, d_t, d_a, d_v, seq_len, bank_capacity, precision, use_graph, seed, device_anchors, batch=batch)
        self.precision = precision
        check(self.lib.mimrl_device_check())
        self.entries, (n_main, n_crit) = _lib.layout_entries(self.cfg)
        f32 = dict(dtype=torch.float32, device=self.device)
        B, T = self.cfg.batch, self.cfg.seq_len
        self.m_anchor = B // self.cfg.k_neighbor
        z = lambda *s: torch.zeros(*s, **f32)
        self.shared = share is not None
        self.main = share.main if share else {k: z(max(n_main, 1)) for k in "pgmv"}
        self.crit = share.crit if share else {k: z(max(n_crit, 1)) for k in "pgmv"}
        if share and (share.main["p"].numel() != max(n_main, 1) or share.crit["p"].numel() != max(n_crit, 1)):
            raise MimrlError("engines that share buckets must have the same parameter layout")
        self.text, self.audio, self.video = z(B, T, d_t), z(B, T, d_a), z(B, T, d_v)
        self.labels = z(B)
        cap = max(int(bank_capacity), 1)
        self.bank = share.bank if share else {"C": z(cap, 1), "F": z(cap, 128), "T": z(cap, 128), "A": z(cap, 128), "V": z(cap, 128)}
        self.anchors = torch.zeros(2, 6, max(self.m_anchor, 1), dtype=torch.int32, device=self.device)
        self.knn_override = torch.zeros(2, 6, max(self.m_anchor * self.cfg.k_neighbor, 1), dtype=torch.int32, device=self.device)
        self._ovr_mask = [0, 0]
        self.bank_c_host = share.bank_c_host if share else None     # host copy of the label bank (exact R^1 kNN tie order)
        self.lr_main = share.lr_main if share else torch.full((1,), float(opt.learning_rate), **f32)
        self.lr_critic = share.lr_critic if share else torch.full((1,), float(opt.learning_rate) * float(opt.mi_lr_rate), **f32)   # Solver.py:140-142
        # [rng step, Adam step (main), Adam step (critics), -]: optimizer / RNG state owned here like m and v
        self.counters = share.counters if share else torch.zeros(4, dtype=torch.int32, device=self.device)
        self.pred = z(B)
        self.feats = z(4, B, 128)
        self.scalars = z(_lib.NSCALARS)
        self.bank_rows = 0
        # named views into the flat buckets (state_dict compatibility)
        self.params: Dict[str, torch.Tensor] = {}
        self.grads: Dict[str, torch.Tensor] = {}
        for name, group, off, shape in self.entries:
            bucket = self.crit if group == 1 else self.main
            n = int(np.prod(shape))
            self.params[name] = bucket["p"][off:off + n].view(*shape)
            self.grads[name] = bucket["g"][off:off + n].view(*shape)
        self.stream = torch.cuda.current_stream(self.device)
        # Handles that share parameter buckets (``share``) each cache bf16 weight images of them: a version counter in the SHARED state
        # says when another handle has stepped or loaded parameters since this one looked, and every compute call checks it (ADVICE r02:
        # a caller alternating sibling handles would otherwise train the bf16 paths on stale weights with no error)
        self._pver = share._pver if share is not None else {"v": 0}
        self._seen = self._pver["v"]
        h = C.c_void_p()
        check(self.lib.mimrl_create(C.byref(self.cfg), C.c_void_p(self.stream.cuda_stream), C.byref(h)))
        self.handle = h
        self._bind()

    def _bind(self):
        b = Buffers()
        b.main_p, b.main_g, b.main_m, b.main_v = (_ptr(self.main[k]) for k in "pgmv")
        b.crit_p, b.crit_g, b.crit_m, b.crit_v = (_ptr(self.crit[k]) for k in "pgmv")
        b.text, b.audio, b.video, b.labels = _ptr(self.text), _ptr(self.audio), _ptr(self.video), _ptr(self.labels)
        b.bank_c, b.bank_f, b.bank_t, b.bank_a, b.bank_v = (_ptr(self.bank[k]) for k in "CFTAV")
        b.anchors = _ptr(self.anchors)
        b.lr_main, b.lr_critic = _ptr(self.lr_main), _ptr(self.lr_critic)
        b.pred, b.feats, b.scalars = _ptr(self.pred), _ptr(self.feats), _ptr(self.scalars)
        b.counters = _ptr(self.counters)
        b.knn_override = _ptr(self.knn_override)
        self._buffers = b
        check(self.lib.mimrl_bind(self.handle, C.byref(b)))

    def close(self):
        if getattr(self, "handle", None):
            torch.cuda.synchronize(self.device)
            self.lib.mimrl_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ state
    def load_params(self, state: Dict[str, "np.ndarray | torch.Tensor"], strict: bool = True):
        for name, view in self.params.items():
            if name not in state:
                if strict:
                    raise KeyError(name)
                continue
            src = state[name]
            src = torch.from_numpy(np.ascontiguousarray(src)) if isinstance(src, np.ndarray) else src
            view.copy_(src.to(torch.float32).reshape(view.shape))
        self.params_changed()

    def params_changed(self):
        """Call after writing parameter tensors from outside the engine (the bf16 weight images are rebuilt lazily)."""
        check(self.lib.mimrl_params_changed(self.handle))
        self._pver["v"] += 1                     # ... and the sibling handles' images are stale as well
        self._seen = self._pver["v"]

    def _coherent(self):
        """Before any compute call: rebuild this handle's cached images if a sibling handle changed the shared parameters."""
        if self._seen != self._pver["v"]:
            check(self.lib.mimrl_params_changed(self.handle))
            self._seen = self._pver["v"]

    def _stepped(self):
        self._pver["v"] += 1
        self._seen = self._pver["v"]

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.detach().clone() for k, v in self.params.items()}

    def set_batch(self, text, audio, video, labels):
        """text: [B,T,d_t] post-BERT features; audio/video: [B,T,d]; labels: [B] or [B,1] (host or device tensors)."""
        for dst, src in ((self.text, text), (self.audio, audio), (self.video, video)):
            src = torch.as_tensor(src)
            dst.copy_(src.reshape(dst.shape), non_blocking=True)
        self.labels.copy_(torch.as_tensor(labels).reshape(-1), non_blocking=True)

    def set_banks(self, C_all, F_all, T_all, A_all, V_all):
        """Previous epoch's stage-2 labels/features (Solver.py:223-227,244).  Empty => epoch-0 rule."""
        n = 0 if C_all is None else len(C_all)
        if n:
            if n > self.cfg.bank_capacity:
                raise MimrlError(f"bank of {n} rows exceeds capacity {self.cfg.bank_capacity}")
            for k, src in zip("CFTAV", (C_all, F_all, T_all, A_all, V_all)):
                src = torch.as_tensor(src)
                self.bank[k][:n].copy_(src.reshape(n, -1), non_blocking=True)
        self.bank_rows = n
        self.bank_c_host = None if not n else np.ascontiguousarray(torch.as_tensor(C_all).detach().cpu().numpy().reshape(-1), dtype=np.float32)
        check(self.lib.mimrl_set_bank_rows(self.handle, n))

    def set_bank_rows(self, n: int):
        """Number of valid rows in the (possibly shared) bank tensors."""
        self.bank_rows = int(n)
        check(self.lib.mimrl_set_bank_rows(self.handle, int(n)))

    def optimizer_state(self) -> Dict[str, torch.Tensor]:
        """Both Adam optimizers (Solver.py:57-62 'optim_main' / 'optim_vmi'): moments as flat buckets + the step counters.  The flat
        buckets are only meaningful together with ``layout_fingerprint()`` (offsets moved in round 5: the layer-0 recurrence tensors went to
        the tail of the main bucket); ``named_moments()`` is the layout-independent form checkpoints store."""
        return {"main_m": self.main["m"].clone(), "main_v": self.main["v"].clone(), "crit_m": self.crit["m"].clone(),
                "crit_v": self.crit["v"].clone(), "counters": self.counters.clone(), "lr_main": self.lr_main.clone(),
                "lr_critic": self.lr_critic.clone()}

    def layout_fingerprint(self) -> str:
        """Hash of the ABI version and every (name, bucket, offset, shape) of the flat-bucket layout (csrc/layout.cpp)."""
        import hashlib
        h = hashlib.sha256(f"abi{int(self.lib.mimrl_abi_version())}".encode())
        for name, group, off, shape in self.entries:
            h.update(f"|{name}:{group}:{off}:{tuple(int(x) for x in shape)}".encode())
        return h.hexdigest()[:32]

    def named_moments(self) -> Dict[str, Dict[str, Dict[str, torch.Tensor]]]:
        """{'main' | 'critic': {'m' | 'v': {parameter name: tensor of the parameter's shape}}} -- Adam's exp_avg / exp_avg_sq per
        parameter, under the reference's state_dict names: independent of where the tensor sits in its bucket."""
        out = {"main": {"m": {}, "v": {}}, "critic": {"m": {}, "v": {}}}
        for name, group, off, shape in self.entries:
            bucket, key = (self.crit, "critic") if group == 1 else (self.main, "main")
            n = int(np.prod(shape))
            for k in "mv":
                out[key][k][name] = bucket[k][off:off + n].view(*shape).clone()
        return out

    def load_named_moments(self, group: str, m: Dict[str, torch.Tensor], v: Dict[str, torch.Tensor]):
        """Inverse of ``named_moments`` for one optimizer ('main' | 'critic'); every parameter of the group must be present."""
        for name, g, off, shape in self.entries:
            if (g == 1) != (group == "critic"):
                continue
            bucket = self.crit if g == 1 else self.main
            n = int(np.prod(shape))
            for k, src in (("m", m), ("v", v)):
                if name not in src:
                    raise KeyError(f"optimizer state of {name!r} missing from the checkpoint")
                bucket[k][off:off + n].copy_(torch.as_tensor(src[name]).to(torch.float32).reshape(-1))

    def load_optimizer_state(self, st: Dict[str, torch.Tensor]):
        for k, dst in (("main_m", self.main["m"]), ("main_v", self.main["v"]), ("crit_m", self.crit["m"]), ("crit_v", self.crit["v"]),
                       ("counters", self.counters), ("lr_main", self.lr_main), ("lr_critic", self.lr_critic)):
            if k in st and st[k] is not None:
                dst.copy_(torch.as_tensor(st[k]).to(dst.dtype).reshape(dst.shape))

    def set_anchors(self, stage: int, anchors, exact_ties: bool = False, bank_c=None):
        """anchors: int array [6, B//k] -- the six ``np.random.choice`` draws of one stage (Model.py:81).
        ``exact_ties``: the two label-conditioned estimators (ta_c, tv_c: kNN in the 1-column label bank, Model.py:327,335)
        get their neighbour rows from the host-side restatement of scikit-learn's KDTree (csrc/knn_r1.cpp) instead of the
        device kernel, so that ties between equal labels -- real MOSI / MOSEI labels are discrete -- resolve exactly as in
        the reference.  (The other four calls search 128-column banks of continuous features: no ties.)"""
        an = np.asarray(anchors, dtype=np.int32).reshape(6, self.m_anchor)
        self.anchors[stage - 1].copy_(torch.from_numpy(an), non_blocking=True)
        mask = 0
        if exact_ties:
            zc = self.bank_c_host if bank_c is None else np.asarray(bank_c, np.float32).reshape(-1)
            if zc is None:
                raise MimrlError("exact_ties needs the label bank on the host (set_banks first)")
            for e in (1, 3):                                     # ta_c, tv_c
                idx = _lib.knn_r1_host(zc[:self.bank_rows], an[e], self.cfg.k_neighbor)
                if idx is not None:
                    self.knn_override[stage - 1, e].copy_(torch.from_numpy(idx.reshape(-1)), non_blocking=True)
                    mask |= 1 << e
        if mask != self._ovr_mask[stage - 1]:
            check(self.lib.mimrl_set_knn_override_mask(self.handle, stage, mask))
            self._ovr_mask[stage - 1] = mask

    def set_lr(self, lr_main: float, lr_critic: float):
        self.lr_main.fill_(float(lr_main))
        self.lr_critic.fill_(float(lr_critic))

    # ------------------------------------------------------------------ compute (all asynchronous)
    def stage1_step(self):
        self._coherent()
        check(self.lib.mimrl_stage1_step(self.handle))
        self._stepped()

    def stage2_step(self):
        self._coherent()
        check(self.lib.mimrl_stage2_step(self.handle))
        self._stepped()

    def set_stage2_prefetch(self, on):
        """Overlap mode of Solver.step(): the stage-2 forward pass runs beside stage 1 (see include/mimrl.h).
        ``on`` = 2 ("deferred tail"): data-parallel variant, the stage-2 forward tail is issued by ``stage2_forward_tail``."""
        check(self.lib.mimrl_set_stage2_prefetch(self.handle, int(on)))

    def stage2_forward_tail(self):
        self._coherent()
        check(self.lib.mimrl_stage2_forward_tail(self.handle))

    def set_grad_scale(self, scale: float):
        """Multiply the gradient buckets by ``scale`` inside the fused clip+Adam (1 / world_size after a SUM all-reduce)."""
        check(self.lib.mimrl_set_grad_scale(self.handle, float(scale)))

    def set_comm(self, unique_id: bytes, world: int, rank: int):
        """Give the engine its own RCCL communicator (include/mimrl.h: mimrl_set_comm; collective over the ranks).  From then on ``step`` /
        ``stage1_step`` / ``stage2_step`` all-reduce the stage's gradient bucket themselves, inside the captured graph.  ``unique_id`` =
        ``HipEngine.comm_unique_id()`` of rank 0, handed to every rank by the caller; ``None`` removes the communicator."""
        self._coherent()
        if unique_id is None:
            check(self.lib.mimrl_set_comm(self.handle, None, 1, 0))
            self.comm_world = 0
            return
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        check(self.lib.mimrl_set_comm(self.handle, C.cast(buf, C.c_void_p), int(world), int(rank)))
        self.comm_world = int(world)
        self.set_grad_scale(1.0 / world)

    def set_comm_critic_bf16(self, on: bool):
        """The critic bucket's all-reduce in bf16 (include/mimrl.h: mimrl_set_comm_critic_bf16): half the bytes of the larger collective."""
        self._coherent()
        check(self.lib.mimrl_set_comm_critic_bf16(self.handle, int(bool(on))))

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_char * 128)()
        check(_lib.load().mimrl_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def step(self):
        """One stage-1 (critics) + one stage-2 (model) update on the bound batch."""
        self._coherent()
        check(self.lib.mimrl_two_stage_step(self.handle))
        self._stepped()

    def stage_grads(self, stage: int):
        self._coherent()
        check(self.lib.mimrl_stage_grads(self.handle, stage))

    def stage_grads_part(self, stage: int, part: int):
        """``stage_grads(2)`` in two launches (include/mimrl.h: mimrl_stage_grads_part): after part 0 every main-bucket gradient except the
        layer-0 GRU tensors is final (``late_grad_ranges``), part 1 finishes those."""
        self._coherent()
        check(self.lib.mimrl_stage_grads_part(self.handle, stage, part))

    def late_grad_ranges(self):
        """[(start, stop)] float ranges of the main gradient bucket that are final only after ``stage_grads_part(2, 1)``: the layer-0
        recurrence tensors (rnn_a / rnn_v ``*_l0*``), which the layout puts at the TAIL of the bucket (round 5) -- one range; everything
        in front of it is final after part 0."""
        off = int(self.lib.mimrl_main_late_offset(self.handle))
        n = self.main["g"].numel()
        return [(off, n)] if off < n else []

    def stage_apply(self, stage: int):
        self._coherent()
        check(self.lib.mimrl_stage_apply(self.handle, stage))
        self._stepped()

    def bucket_grad(self, stage: int) -> torch.Tensor:
        """Flat gradient bucket updated by ``stage`` (1: critics, 2: main model) -- the all-reduce payload."""
        return self.crit["g"] if stage == 1 else self.main["g"]

    def has_update(self, stage: int) -> bool:
        return stage == 2 or self.bank_rows > 0          # epoch-0 rule: stage 1 does nothing without banks

    def forward(self, train: bool = False, with_losses: bool = False):
        self._coherent()
        check(self.lib.mimrl_forward(self.handle, int(train), int(with_losses)))

    def estimate(self, stage: int):
        """Estimators only, on the features left by the last forward (Model.compute_vmi_loss_stage1/2)."""
        self._coherent()
        check(self.lib.mimrl_estimate(self.handle, stage))

    # ------------------------------------------------------------------ test probes (include/mimrl.h: mimrl_probe_*)
    def probe_cube(self, x, dout=None):
        """The CubeMLP stack through the engine's own kernels (current precision mode).  x [B,L,3,128] -> out; with ``dout`` also
        -> (out, dx) and every ``mlp_encoder.*`` gradient in ``self.grads`` (the main bucket is zeroed first)."""
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device).contiguous()
        c = self.cfg
        nb = c.n_blocks
        out = torch.empty(c.batch, c.d_outs[nb - 1][0], c.d_outs[nb - 1][1], c.d_outs[nb - 1][2], dtype=torch.float32, device=self.device)
        if dout is None:
            check(self.lib.mimrl_probe_cube(self.handle, _ptr(x), _ptr(out), None, None))
            return out
        dout = torch.as_tensor(dout, dtype=torch.float32, device=self.device).contiguous().reshape(out.shape)
        dx = torch.empty_like(x)
        check(self.lib.mimrl_probe_cube(self.handle, _ptr(x), _ptr(out), _ptr(dout), _ptr(dx)))
        return out, dx

    def probe_mi(self, stage: int, feats):
        """The five MI estimators of ``stage`` forward + backward on ``feats`` [4,B,128] (F,T,A,V) through the engine's own kernels.
        -> dict(mi [5], mi_loss [5], scores [5,B,B] (concat critic) or None, dtin [5,2,B,128] (stage 2) or None); stage 1 leaves
        the ``vmi_estimator_*`` gradients in ``self.grads``."""
        self.feats.copy_(torch.as_tensor(feats, dtype=torch.float32).reshape(self.feats.shape))
        B = self.cfg.batch
        mi = torch.empty(2, 5, dtype=torch.float32, device=self.device)
        scores = torch.empty(5, B, B, dtype=torch.float32, device=self.device) if self.cfg.critic_type == 1 else None
        dtin = torch.empty(5, 2, B, 128, dtype=torch.float32, device=self.device) if stage == 2 else None
        check(self.lib.mimrl_probe_mi(self.handle, stage, _ptr(mi), _ptr(scores), _ptr(dtin)))
        return {"mi": mi[0], "mi_loss": mi[1], "scores": scores, "dtin": dtin}

    def probe_cmi(self, stage: int, cmi_in):
        """The six CMI classifiers of ``stage`` forward + loss + backward on a caller-assembled batch [6, 2n, 384] through the engine's
        own kernels.  -> dict(logits [6,2n,2], bce [6], cmi [6], dcin [6,2n,384] (stage 2) or None); stage 1 leaves the
        ``vcmi_estimator_*`` gradients in ``self.grads``."""
        n = self.m_anchor * self.cfg.k_neighbor
        x = torch.as_tensor(cmi_in, dtype=torch.float32, device=self.device).contiguous().reshape(6, 2 * n, 384)
        logits = torch.empty(6, 2 * n, 2, dtype=torch.float32, device=self.device)
        vals = torch.empty(2, 6, dtype=torch.float32, device=self.device)
        dcin = torch.empty(6, 2 * n, 384, dtype=torch.float32, device=self.device) if stage == 2 else None
        check(self.lib.mimrl_probe_cmi(self.handle, stage, _ptr(x), _ptr(logits), _ptr(vals), _ptr(dcin)))
        return {"logits": logits, "bce": vals[0], "cmi": vals[1], "dcin": dcin}

    def probe_encoders(self, dcube=None, dmean=None):
        """Model.forward's encoders on the bound batch through the engine's own kernels (include/mimrl.h: mimrl_probe_encoders):
        -> cube_x [B,L,3,128]; with ``dcube`` (and optionally ``dmean`` [3,B,128]) every W_t / rnn_* / ln_* gradient lands in ``self.grads``."""
        c = self.cfg
        x = torch.empty(c.batch, c.time_len, 3, 128, dtype=torch.float32, device=self.device)
        if dcube is None:
            check(self.lib.mimrl_probe_encoders(self.handle, _ptr(x), None, None))
            return x
        dcube = torch.as_tensor(dcube, dtype=torch.float32, device=self.device).contiguous().reshape(x.shape)
        dm = None if dmean is None else torch.as_tensor(dmean, dtype=torch.float32, device=self.device).contiguous().reshape(3, c.batch, 128)
        check(self.lib.mimrl_probe_encoders(self.handle, _ptr(x), _ptr(dcube), _ptr(dm)))
        self._keep = (dcube, dm)          # alive until the asynchronous copies have run
        return x

    def probe_knn(self, stage: int) -> torch.Tensor:
        """Neighbour rows [6, B//k, k] of the last kNN product sample of ``stage`` (include/mimrl.h: mimrl_probe_knn)."""
        out = torch.empty(6, self.m_anchor, self.cfg.k_neighbor, dtype=torch.int32, device=self.device)
        check(self.lib.mimrl_probe_knn(self.handle, stage, _ptr(out)))
        return out

    STAMP_IDS = ("gru_fwd_l0", "gru_fwd_l1", "gru_bwd_l1", "gru_bwd_l0")

    def kernel_stamps(self, slots: int = 1 << 14):
        """Switch the in-kernel launch stamps of the recurrence kernels on (include/mimrl.h: mimrl_set_kernel_stamps) and clear the
        ring.  ``slots`` (power of two) >= 2 x the steps between two reads."""
        if getattr(self, "_stamps", None) is None or self._stamps.shape[0] != slots:
            self._stamps = torch.empty(slots, 4, 2, dtype=torch.int64, device=self.device)
            check(self.lib.mimrl_set_kernel_stamps(self.handle, _ptr(self._stamps), slots))
        self._stamps.fill_(-1)

    def read_kernel_stamps(self) -> Dict[str, "np.ndarray"]:
        """-> {kernel: launch durations in microseconds} of every launch stamped since ``kernel_stamps()`` (synchronises)."""
        r = self._stamps.cpu().numpy().view(np.uint64)
        out = {}
        for i, name in enumerate(self.STAMP_IDS):
            t0, t1 = r[:, i, 0], ~r[:, i, 1]
            ok = (r[:, i, 0] != np.uint64(0xFFFFFFFFFFFFFFFF)) & (r[:, i, 1] != np.uint64(0xFFFFFFFFFFFFFFFF))
            out[name] = (t1[ok].astype(np.float64) - t0[ok].astype(np.float64)) / 100.0     # 100 MHz ticks -> us
        return out

    def profile(self, on: bool):
        check(self.lib.mimrl_profile_enable(self.handle, int(on)))

    def profile_read(self):
        """-> {phase: (total_ms, launches)} since the last read (synchronises)."""
        n = len(_lib.PHASES)
        ms, cnt = (C.c_float * n)(), (C.c_int32 * n)()
        check(self.lib.mimrl_profile_read(self.handle, ms, cnt))
        return {p: (float(ms[i]), int(cnt[i])) for i, p in enumerate(_lib.PHASES)}

    def profile_read_gemm(self):
        """GEMM family of the eager steps since the last read -> dict(flops, bytes, ms, launches) (synchronises)."""
        out = (C.c_double * 4)()
        check(self.lib.mimrl_profile_read_gemm(self.handle, out))
        return {"flops": out[0], "bytes": out[1], "ms": out[2], "launches": int(out[3])}

    # ------------------------------------------------------------------ overlapped batch upload (double-buffered inputs)
    def _ensure_sets(self):
        """The second input set + the upload machinery (copy stream, events), created on first use."""
        if hasattr(self, "_sets"):
            return
        self._sets = [(self.text, self.audio, self.video, self.labels),
                      tuple(torch.empty_like(t) for t in (self.text, self.audio, self.video, self.labels))]
        self._active = 0
        # high priority = its own hardware queue: on a normal-priority stream the copy's barrier packet shares one of the 4
        # hardware queues with a branch of the step graph and the step starts BEHIND the upload (cfg2: 1.47 instead of 1.09 ms)
        self._upload_legacy = os.environ.get("MIMRL_UPLOAD_LEGACY") is not None   # A/B knob: normal-priority stream + stream wait
        self._copy_stream = torch.cuda.Stream(self.device, priority=0 if self._upload_legacy else -1)
        self._staged_ev = torch.cuda.Event()
        self._free_ev = [torch.cuda.Event(), torch.cuda.Event()]   # set q is no longer read by the device after this point
        for ev in self._free_ev:
            ev.record(self.stream)

    def stage_batch(self, text, audio, video, labels):
        """Start the host->device copy of the NEXT batch into the IDLE input set on a copy stream (it overlaps the step that
        is running on the active set).  Sources in pinned memory make the copy truly asynchronous."""
        self._ensure_sets()
        idle = 1 - self._active
        # steps that read the idle set have finished.  A HOST wait (the caller runs at most one step ahead of the device), not
        # hipStreamWaitEvent: a copy parked behind a not-yet-complete event of the compute stream cost the step 0.6 ms on this
        # runtime (tools/fresh_variants.py: 1.63 ms against 1.09 with the host wait)
        if not self._upload_legacy:
            self._free_ev[idle].synchronize()
        with torch.cuda.stream(self._copy_stream):
            if self._upload_legacy:
                self._copy_stream.wait_event(self._free_ev[idle])
            for dst, src in zip(self._sets[idle], (text, audio, video, labels)):
                dst.copy_(torch.as_tensor(src).reshape(dst.shape), non_blocking=True)
            self._staged_ev.record(self._copy_stream)

    def commit_batch(self):
        """Make the staged batch the bound one: switch the engine to the other input set (mimrl_set_inputs: host-only, the
        graphs are cached per set) -- no device copy."""
        old = self._active
        new = 1 - old
        self._free_ev[old].record(self.stream)                         # everything enqueued so far may still read the old set
        self.stream.wait_event(self._staged_ev)
        t, a, v, y = self._sets[new]
        check(self.lib.mimrl_set_inputs(self.handle, new, _ptr(t), _ptr(a), _ptr(v), _ptr(y)))
        self.text, self.audio, self.video, self.labels = t, a, v, y
        self._active = new

    # ------------------------------------------------------------------ epoch-ordered critic pass with look-ahead forward (round 6)
    def stage1_pass(self, batches, on_step=None):
        """One pass of critic updates (reference: the inner loop of Solver.py:200-216) over ``batches`` = an iterable of
        ``(text, audio, video, labels)`` DEVICE tensors of this engine's batch size.  The main model is frozen in such a pass, so
        Model.forward of batch i + 1 runs beside the estimators / clip / Adam of batch i (`mimrl_stage1_pipe`): the next batch is copied
        into the idle input set (device to device, on the engine's stream) in front of the step on the current one.  ``on_step(self)`` is
        called behind every enqueued step (the caller accumulates ``self.scalars`` there).  Same losses, dropout masks and anchor draws as
        ``stage1_step`` batch by batch."""
        self._ensure_sets()
        it = iter(batches)
        cur = next(it, None)
        if cur is None:
            return 0
        def put(slot, b):
            for dst, src in zip(self._sets[slot], b):
                dst.copy_(torch.as_tensor(src).reshape(dst.shape), non_blocking=True)
        def bind(slot):
            t, a, v, y = self._sets[slot]
            check(self.lib.mimrl_set_inputs(self.handle, slot, _ptr(t), _ptr(a), _ptr(v), _ptr(y)))
            self.text, self.audio, self.video, self.labels = t, a, v, y
            self._active = slot
        self._coherent()
        with torch.cuda.stream(self.stream):
            bind(1 - self._active); bind(1 - self._active)           # (both sets known to the library; back on the active one)
            put(self._active, cur)
            check(self.lib.mimrl_stage1_pipe_prime(self.handle))
            n = 0
            while cur is not None:
                nxt = next(it, None)
                if nxt is not None:
                    put(1 - self._active, nxt)
                check(self.lib.mimrl_stage1_pipe(self.handle, 1 if nxt is not None else 0))
                self._stepped()
                n += 1
                if on_step is not None:
                    on_step(self)
                if nxt is not None:
                    bind(1 - self._active)
                cur = nxt
            for ev in self._free_ev:                                   # (what was enqueued reads both sets: a later upload waits for it)
                ev.record(self.stream)
        return n

    def read_scalars(self) -> np.ndarray:
        """One device->host read-back (the reference does >= 10 ``.item()`` syncs per iteration)."""
        return self.scalars.detach().cpu().numpy()

    def workspace_bytes(self) -> int:
        return int(self.lib.mimrl_workspace_bytes(self.handle))
