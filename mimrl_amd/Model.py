"""``Model`` with the reference's constructor / forward / compute_vmi_loss_stage{1,2} signatures (Model.py:228-519),
backed by the HIP engine.  Parameters, gradients and Adam moments live in two flat device buckets owned by torch
tensors; ``named_parameters()`` / ``state_dict()`` expose the reference's names as views, so the optimiser split by
substring ('vmi' / 'vcmi' / else; Solver.py:124-133) and checkpointing keep working."""
from __future__ import annotations

from typing import Dict, Iterator, List, Tuple

import numpy as np
import torch

from . import _lib, synth
from .engine import HipEngine


def _rebuild_model(opt, d_t, d_a, d_v, kw, state, training):
    m = Model(opt, d_t, d_a, d_v, **kw)
    m.load_state_dict(state)
    return m.train(training)


class Model:
    def __init__(self, opt, d_t: int, d_a: int, d_v: int, bank_capacity: int = 0, precision: str = None,
                 use_graph: bool = None, device_anchors: bool = None, seq_len: int = None, init: str = "default", rank: int = 0):
        self.opt = opt
        self.d_t, self.d_a, self.d_v, self.d_common = d_t, d_a, d_v, opt.d_common
        assert opt.encoders in ["lstm", "gru", "conv"]                                   # Model.py:237
        assert opt.features_compose_t in ["mean", "cat", "sum"] and opt.features_compose_k in ["mean", "cat", "sum"]
        precision = precision or getattr(opt, "precision", "fp32")
        use_graph = (not getattr(opt, "no_graph", False)) if use_graph is None else use_graph
        device_anchors = (not getattr(opt, "host_anchors", False)) if device_anchors is None else device_anchors
        self.engine = HipEngine(opt, d_t, d_a, d_v, seq_len=seq_len, bank_capacity=bank_capacity, precision=precision,
                                use_graph=use_graph, seed=int(getattr(opt, "seed", 0)) + 7919 * int(rank),   # dropout / anchor streams differ per rank
                                device_anchors=device_anchors)
        self.training = True
        self.k_neighbor = opt.k_neighbor
        shapes = [(n, tuple(v.shape)) for n, v in self.engine.params.items()]
        gen = synth.default_state if init == "default" else synth.portable_state
        self.engine.load_params(gen(shapes, int(getattr(opt, "seed", 0))))

    # ---- nn.Module-like surface -------------------------------------------------------------
    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def cuda(self, *a, **k):
        return self

    def to(self, *a, **k):
        """nn.Module.to: the engine lives on the GPU it was created on; dtype / device moves do not apply (fails loudly on a CPU target)."""
        tgt = [x for x in list(a) + list(k.values()) if isinstance(x, (str, torch.device))]
        if any(torch.device(t).type == "cpu" for t in tgt):
            raise _lib.MimrlError("Model.to('cpu'): this model has no CPU path (oracle/ is test infrastructure)")
        return self

    def zero_grad(self, set_to_none: bool = False):
        """The fused clip+Adam leaves both gradient buckets zeroed; this is for callers that run stage_grads without stage_apply."""
        self.engine.main["g"].zero_()
        self.engine.crit["g"].zero_()

    def __reduce__(self):
        """torch.save(model) / pickle: the engine handle cannot travel; what is saved is how to rebuild it + the parameters
        (optimizer state is the checkpoint's business: Solver.checkpoint)."""
        c = self.engine.cfg
        kw = dict(bank_capacity=int(c.bank_capacity), precision=self.engine.precision, use_graph=bool(c.use_graph),
                  device_anchors=bool(c.device_anchors), seq_len=int(c.seq_len))
        return (_rebuild_model, (self.opt, self.d_t, self.d_a, self.d_v, kw, {k: v.detach().cpu() for k, v in self.state_dict().items()},
                                 self.training))

    @property
    def module(self):          # the reference reaches through nn.DataParallel (Customization.py:99,107)
        return self

    def named_parameters(self) -> Iterator[Tuple[str, torch.Tensor]]:
        return iter(self.engine.params.items())

    def parameters(self):
        return iter(self.engine.params.values())

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return self.engine.state_dict()

    def load_state_dict(self, state, strict: bool = True):
        self.engine.load_params(state, strict=strict)

    # ---- Model.forward (Model.py:388-519) ---------------------------------------------------------
    def forward(self, bert_sentences, bert_sentence_types, bert_sentence_att_mask, a, v, return_features=False, labels=None):
        """``bert_sentences`` carries the BERT last-hidden-state features [B,T,d_t] (BERT is outside the hot path)."""
        e = self.engine
        for dst, src in ((e.text, bert_sentences), (e.audio, a), (e.video, v)):
            dst.copy_(torch.as_tensor(src).reshape(dst.shape), non_blocking=True)
        if labels is not None:
            e.labels.copy_(torch.as_tensor(labels).reshape(-1), non_blocking=True)
        e.forward(train=self.training, with_losses=False)
        out = e.pred.reshape(-1, 1)
        return [out, e.feats[0], e.feats[1], e.feats[2], e.feats[3]] if return_features else [out]

    __call__ = forward

    # ---- Model.compute_vmi_loss_stage1/2 (Model.py:305-386) ---------------------------------------
    def _estimate(self, stage, labels, banks, predictions=None, feats=None):
        e = self.engine
        # The engine estimates on ITS feature / prediction buffers (mimrl_buffers.feats / .pred).  Model.forward returns views of them,
        # so the usual call passes them straight back; tensors that are NOT those views (a caller's own features) are copied in, so that
        # the arguments mean what they mean in the reference (Model.py:305, 343) instead of being silently ignored
        if feats is not None:
            for i, f in enumerate(feats):
                f = torch.as_tensor(f)
                if not (f.is_cuda and f.data_ptr() == e.feats[i].data_ptr()):
                    e.feats[i].copy_(f.reshape(e.feats[i].shape), non_blocking=True)
        if predictions is not None:
            pr = torch.as_tensor(predictions)
            if not (pr.is_cuda and pr.data_ptr() == e.pred.data_ptr()):
                e.pred.copy_(pr.reshape(-1), non_blocking=True)
        e.labels.copy_(torch.as_tensor(labels).reshape(-1), non_blocking=True)
        if banks is not None:
            e.set_banks(*banks)
        if not e.cfg.device_anchors:
            e.set_anchors(stage, synth.draw_anchors(e.bank_rows, e.m_anchor, 6), exact_ties=True)   # consumes numpy's global RNG like Model.py:81
        e.estimate(stage)

    def compute_vmi_loss_stage1(self, predictions, labels, F_F, T_F, A_F, V_F, C_F_all, F_F_all, T_F_all, A_F_all, V_F_all):
        """Model.py:305-341 on the given features (views of the last forward's buffers are used in place, other tensors are copied
        into them).  -> (11 mis, 11 losses)"""
        self._estimate(1, labels, (C_F_all, F_F_all, T_F_all, A_F_all, V_F_all), predictions, (F_F, T_F, A_F, V_F))
        s = self.engine.scalars
        return list(s[_lib.S1_MIS:_lib.S1_MIS + 11]), list(s[_lib.S1_LOSSES:_lib.S1_LOSSES + 11])

    def compute_vmi_loss_stage2(self, predictions, labels, F_F, T_F, A_F, V_F, C_F_all, F_F_all, T_F_all, A_F_all, V_F_all):
        self._estimate(2, labels, (C_F_all, F_F_all, T_F_all, A_F_all, V_F_all), predictions, (F_F, T_F, A_F, V_F))
        s = self.engine.scalars
        return list(s[_lib.S2_MIS:_lib.S2_MIS + 8]), list(s[_lib.S2_LOSSES:_lib.S2_LOSSES + 8])
