"""``Solver`` with the reference's surface (Solver.py:18-531) re-expressed around two fused device steps.

New (named by BASELINE.json's north star, absent in the reference): ``Solver.step(datas)`` = one stage-1 update
(body of Solver.py:205-214) + one stage-2 update (body of :221-236) on one batch.  ``train()`` keeps the reference's
epoch ordering -- ``stage1_n`` full critic passes (skipped in epoch 0), then one model pass that also builds the next
epoch's feature banks -- but accumulates losses / MI terms / banks on the device and reads back once per epoch
(the reference performs >= 10 ``.item()`` syncs and 18 bank D2H copies per iteration).

Batches shorter than ``--batch_size`` (the last batch of a loader with the reference's default ``drop_last=False``,
Parameters.py:21) run on a second engine handle of that batch size bound to the SAME parameter / Adam buckets, step
counters and banks (`HipEngine(share=...)`), i.e. the same optimizer, as in the reference.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib, dist as mdist, synth
from .Customization import compute_custumized_loss, compute_outputs_from_model, other_model_operations
from .engine import HipEngine
from .Model import Model
from .Utils import log_message, set_logger


class _Plateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau with its defaults (threshold 1e-4 'rel', cooldown 0, min_lr 0, eps 1e-8), as the
    reference builds it for `--lr_decrease plateau` (Solver.py:163-166: mode 'min' for regression else 'max', patience
    `--lr_decrease_iter`, factor `--lr_decrease_rate`), one instance per optimizer group; stepped with the epoch's validation loss
    (Solver.py:50-52).  The rates live in the engine's fused Adam, so the schedule is host arithmetic on two floats."""

    def __init__(self, lr, mode, patience, factor, threshold=1e-4, eps=1e-8):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        self.lr, self.mode, self.patience, self.factor, self.threshold, self.eps = float(lr), mode, int(patience), float(factor), threshold, eps
        self.best = float("inf") if mode == "min" else -float("inf")
        self.num_bad_epochs = 0

    def is_better(self, a):
        return a < self.best * (1.0 - self.threshold) if self.mode == "min" else a > self.best * (self.threshold + 1.0)

    def step(self, metric):
        current = float(metric)
        if self.is_better(current):
            self.best, self.num_bad_epochs = current, 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            new_lr = max(self.lr * self.factor, 0.0)
            if self.lr - new_lr > self.eps:
                self.lr = new_lr
            self.num_bad_epochs = 0
        return self.lr


def _loader_samples(loader, batch_size):
    if hasattr(loader, "num_samples"):
        return int(loader.num_samples())
    if isinstance(loader, (list, tuple)):
        return int(sum(len(d[5]) for d in loader))
    return len(loader) * batch_size


class Solver:
    def __init__(self, opt, loaders=None):
        self.opt = opt
        self.world, self.rank, self.local_rank = mdist.init_from_env()
        if opt.loss != "MAE":
            raise NotImplementedError(f"--loss {opt.loss}: only MAE is on the MI355X hot path (SURVEY.md section 2)")
        if opt.optm != "Adam":
            raise NotImplementedError(f"--optm {opt.optm}: only Adam is fused on the MI355X hot path")
        if torch.cuda.is_available():   # BEFORE the loaders: a resident dataset lands on the current device (data.py)
            torch.cuda.set_device(self.local_rank % torch.cuda.device_count())   # (more ranks than GPUs: ranks share devices)
        if loaders is None:
            from .data import get_data_loader
            loaders = get_data_loader(opt, self.rank, self.world)          # training data sharded by rank
        self.train_loader, self.valid_loader, self.test_loader, self.d_t, self.d_a, self.d_v = loaders
        # banks hold one row per training sample of EVERY rank (all-gathered once per epoch)
        cap = _loader_samples(self.train_loader, opt.batch_size) * self.world
        self.model = Model(opt, self.d_t, self.d_a, self.d_v, bank_capacity=cap, rank=self.rank)
        other_model_operations(self.model, opt)
        self.engine = self.model.engine
        self._tails = {}                 # batch size -> engine sharing self.engine's buckets (partial last batch)
        self._active = self.engine
        self._want_prefetch = False
        if self.world > 1:      # identical replicas: rank 0's initial parameters everywhere
            mdist.broadcast_(self.engine.main["p"])
            mdist.broadcast_(self.engine.crit["p"])
            self.engine.params_changed()
            # the engine's own RCCL communicator: the gradient all-reduces become part of its captured graphs (dist.attach_comm); on a
            # process group that is not RCCL (the gloo tests) the torch.distributed transport of dist.ddp_* stays in charge
            mdist.attach_comm(self.engine, self.world, self.rank)
        self.base_lr = float(opt.learning_rate)
        self.epoch = 0
        self._plateau = None
        if opt.lr_decrease == "plateau":                                                   # Solver.py:163-166
            mode = "min" if getattr(opt, "task", "regression") == "regression" else "max"
            self._plateau = [_Plateau(lr, mode, int(opt.lr_decrease_iter), float(opt.lr_decrease_rate))
                             for lr in (self.base_lr, self.base_lr * float(opt.mi_lr_rate))]
        self.task_path = os.path.join("./TaskRuning", str(opt.task_name))                  # Solver.py:107-112
        self.best_valid_model_path = os.path.join(self.task_path, "best_valid_model.pth.tar")
        self.best_test_model_path = os.path.join(self.task_path, "best_test_model.pth.tar")

    # ------------------------------------------------------------------ learning-rate schedules (Solver.py:153-169)
    def lr_factor(self, epoch: int) -> float:
        o = self.opt
        if o.lr_decrease == "step":
            return o.lr_decrease_rate ** (epoch // int(o.lr_decrease_iter))
        if o.lr_decrease == "multi_step":
            ms = [int(x) for x in str(o.lr_decrease_iter).split("-")]
            return o.lr_decrease_rate ** sum(epoch >= m for m in ms)
        if o.lr_decrease == "exp":
            return o.lr_decrease_rate ** epoch
        if o.lr_decrease == "plateau":
            return self._plateau[0].lr / self.base_lr
        raise NotImplementedError(f"--lr_decrease {o.lr_decrease}")

    def _apply_lr(self, epoch: int):
        if self._plateau is not None:            # metric-driven: the rates move in lr_schedule_step, not with the epoch number
            self.engine.set_lr(self._plateau[0].lr, self._plateau[1].lr)
            return
        f = self.lr_factor(epoch)
        self.engine.set_lr(self.base_lr * f, self.base_lr * float(self.opt.mi_lr_rate) * f)

    def lr_schedule_step(self, val_loss=None):
        """End-of-epoch scheduler step (Solver.py:49-55).  step / multi_step / exp are closed forms of the epoch number (`lr_factor`);
        'plateau' feeds the validation loss to both groups' ReduceLROnPlateau."""
        if self._plateau is not None:
            for s in self._plateau:
                s.step(val_loss)

    # ------------------------------------------------------------------ engines
    def get_label_from_datas(self, datas):
        return datas[5]                                                       # Solver.py:273-275 ('Dec' layout)

    def _engine_for(self, B: int, training: bool = True) -> HipEngine:
        """The engine whose batch size is ``B``: the primary one, or a lazily created sibling sharing its optimizer."""
        B = int(B)
        e = self.engine if B == self.engine.cfg.batch else self._tails.get(B)
        if e is None:
            if B > self.engine.cfg.batch or B < int(self.opt.k_neighbor):
                raise _lib.MimrlError(f"batch of {B} rows: need k_neighbor <= B <= --batch_size {self.engine.cfg.batch}")
            if self.world > 1 and training:     # (evaluation has no collectives: a short last batch is fine there)
                raise _lib.MimrlError("data-parallel ranks need full training batches (sharded loaders drop the ragged tail)")
            p = self.engine
            e = HipEngine(self.opt, self.d_t, self.d_a, self.d_v, seq_len=p.cfg.seq_len, precision=p.precision,
                          use_graph=bool(p.cfg.use_graph), seed=int(p.cfg.seed), device=p.device,
                          device_anchors=bool(p.cfg.device_anchors), share=p, batch=B)
            self._tails[B] = e
        if e is not self._active:
            e.params_changed()                      # its bf16 weight images went stale while another handle stepped
            self._active = e
        if e.bank_rows != self.engine.bank_rows:
            e.set_bank_rows(self.engine.bank_rows)
        want = (1 if mdist.has_comm(e, self.world) else mdist.ddp_prefetch_mode(self.world)) if self._want_prefetch else 0
        if getattr(e, "_prefetch_on", 0) != want:
            e.set_stage2_prefetch(want)
            e._prefetch_on = want
        return e

    def _load(self, datas) -> HipEngine:
        _, a, v, _, _, labels, feats, _, _, _, _ = datas
        e = self._engine_for(len(labels))
        e.set_batch(feats, a, v, labels)
        return e

    def _iter_loaded(self, loader, training: bool = True):
        """Iterate ``(engine, datas)`` over a loader with each batch bound to its engine, uploading batch i+1 on a copy
        stream while the caller's step on batch i runs (HipEngine.stage_batch / commit_batch)."""
        it = iter(loader)
        cur = next(it, None)
        staged = None
        while cur is not None:
            _, a, v, _, _, labels, feats, _, _, _, _ = cur
            e = self._engine_for(len(labels), training)
            if staged is e:
                e.commit_batch()
            else:
                e.set_batch(feats, a, v, labels)
            nxt = next(it, None)
            yield e, cur
            # the upload of batch i+1 is enqueued AFTER the caller has enqueued its step on batch i: stage_batch waits (on the host)
            # until the step before that one has released the idle input set, and with this order the device already has batch i's
            # step queued while the host waits -- the copy then starts somewhere inside that step, not together with it
            staged = None
            if nxt is not None and len(nxt[5]) == e.cfg.batch:
                e.stage_batch(nxt[6], nxt[1], nxt[2], nxt[5])
                staged = e
            cur = nxt

    def _anchors(self, e, stage):
        if e.bank_rows > 0 and not e.cfg.device_anchors:
            e.set_anchors(stage, synth.draw_anchors(e.bank_rows, e.m_anchor, 6), exact_ties=True, bank_c=self.engine.bank_c_host)

    def _set_banks(self, C_F_all, F_F_all, T_F_all, A_F_all, V_F_all):
        self.engine.set_banks(C_F_all, F_F_all, T_F_all, A_F_all, V_F_all)

    def _prefetch(self, on):
        self._want_prefetch = bool(on)

    # ------------------------------------------------------------------ the hot path
    def _stage(self, e, stage):
        if self.world > 1:
            mdist.ddp_stage_step(e, stage, self.world)
        elif stage == 1:
            e.stage1_step()
        else:
            e.stage2_step()

    def _stage1_pass_pipelined(self, acc) -> bool:
        """One critic pass over ``self.train_loader`` with the NEXT batch's forward pass beside each update (HipEngine.stage1_pass): the main
        model is frozen in such a pass (Solver.py:204-216), so Model.forward(batch i + 1) does not depend on the critic update on batch i.
        Only when every batch is device-resident, full-size and the anchors are drawn on the device (no per-batch host work), single
        process; otherwise False and the caller runs the sequential pass.  Same numbers as the sequential pass, batch by batch.
        OPT-IN (MIMRL_EPOCH_PIPE=1): measured on cfg2 it LOSES to the sequential pass (1.40-1.56 vs 1.25 ms per stage-1 + stage-2 pair) --
        the kNN sampler's merge kernel takes 132-151 us instead of 18 whenever it runs beside the estimators' or the forward pass's kernels
        (DESIGN.md section 7), which eats the overlap."""
        e = self.engine
        if self.world > 1 or not e.cfg.use_graph or not e.cfg.device_anchors or os.environ.get("MIMRL_EPOCH_PIPE", "0") in ("", "0"):
            return False
        B = e.cfg.batch
        batches = []
        for d in self.train_loader:
            if len(d[5]) != B or not all(torch.is_tensor(d[k]) and d[k].is_cuda for k in (6, 1, 2, 5)):
                return False
            batches.append((d[6], d[1], d[2], d[5]))
        if not batches:
            return False
        e = self._engine_for(B)
        def on_step(eng):
            acc[_lib.S1_LOSS] += eng.scalars[_lib.S1_LOSS]
        e.stage1_pass(batches, on_step)
        return True

    def stage1_step(self, datas=None, draw_anchors=True):
        """Critic update (Solver.py:205-214).  Returns the stage-1 loss as a device scalar."""
        self._prefetch(False)                       # a lone stage call is the sequential schedule
        e = self._load(datas) if datas is not None else self._engine_for(self._active.cfg.batch)
        if draw_anchors:
            self._anchors(e, 1)
        self._stage(e, 1)
        return e.scalars[_lib.S1_LOSS]

    def stage2_step(self, datas=None, draw_anchors=True):
        """Model update (Solver.py:221-236).  Returns (loss, mis[8], pred[B,1]) as device tensors."""
        self._prefetch(False)
        e = self._load(datas) if datas is not None else self._engine_for(self._active.cfg.batch)
        if draw_anchors:
            self._anchors(e, 2)
        self._stage(e, 2)
        s = e.scalars
        return s[_lib.S2_LOSS], s[_lib.S2_MIS:_lib.S2_MIS + 8], e.pred.reshape(-1, 1)

    def step(self, datas):
        """One two-stage iteration on one batch: the unit BASELINE.json's metric counts.  Both stages see the same
        batch and stage 1 leaves the main model untouched, so the engine runs the stage-2 forward pass beside stage 1
        (`mimrl_set_stage2_prefetch`); the numbers are those of the sequential order."""
        self._prefetch(True)
        e = self._load(datas)
        self._anchors(e, 1)       # host-drawn anchors (if any) for BOTH stages go up before stage 1: in overlap mode the
        self._anchors(e, 2)       # stage-2 kNN sampler already runs beside stage 1 (same draw order as the reference)
        if self.world > 1:
            mdist.ddp_two_stage_step(e, self.world)
        else:
            e.step()              # mimrl_two_stage_step: both stages as one captured graph
        sc = e.scalars
        return sc[_lib.S1_LOSS], sc[_lib.S2_LOSS], sc[_lib.S2_MIS:_lib.S2_MIS + 8], e.pred.reshape(-1, 1)

    # ------------------------------------------------------------------ Solver.train (Solver.py:194-248)
    def train(self, epoch, train_loader, C_F_all, F_F_all, T_F_all, A_F_all, V_F_all):
        self.model.train()
        self._prefetch(False)                                                  # epoch-ordered passes: stages see different batches
        self._apply_lr(epoch)
        self._set_banks(C_F_all, F_F_all, T_F_all, A_F_all, V_F_all)
        dev = self.engine.device
        acc = torch.zeros(_lib.NSCALARS, device=dev)
        nb = len(train_loader)
        if epoch > 0 and len(C_F_all) > 0:                                     # Solver.py:200-203: epoch 0 skips stage 1
            for _ in range(self.opt.stage1_n):
                if self._stage1_pass_pipelined(acc):
                    continue
                for e, datas in self._iter_loaded(self.train_loader):
                    self._anchors(e, 1)
                    self._stage(e, 1)
                    acc[_lib.S1_LOSS] += e.scalars[_lib.S1_LOSS]
        new = {k: [] for k in "CFTAV"}
        preds, targs = [], []
        for e, datas in self._iter_loaded(train_loader):
            self._anchors(e, 2)
            self._stage(e, 2)
            new["C"].append(e.labels.reshape(-1, 1).clone())                   # Solver.py:223-227 (features of THIS pass)
            for i, k in enumerate("FTAV"):
                new[k].append(e.feats[i].clone())
            preds.append(e.pred.clone())
            targs.append(e.labels.clone())
            acc[32:] += e.scalars[32:]
        banks = [torch.cat(new[k], 0) for k in "CFTAV"]
        if self.world > 1:                                                    # banks are replicated
            banks = [mdist.allgather_rows(x, self.world) for x in banks]
        a = acc.cpu().numpy()                                                 # the ONE read-back of the epoch
        predictions, targets = torch.cat(preds).cpu().numpy().reshape(-1, 1), torch.cat(targs).cpu().numpy().reshape(-1, 1)
        train_score = self.get_score_from_result(predictions, targets)
        mis = [float(x) / nb for x in a[_lib.S2_MIS:_lib.S2_MIS + 8]]
        return (float(a[_lib.S2_LOSS]) / nb, float(a[_lib.S1_LOSS]) / nb, mis, train_score, *banks)

    # ------------------------------------------------------------------ Solver.evaluate (Solver.py:250-270)
    def evaluate(self, valid_loader, C_F_all, F_F_all, T_F_all, A_F_all, V_F_all):
        self.model.eval()
        self._prefetch(False)
        self._set_banks(C_F_all, F_F_all, T_F_all, A_F_all, V_F_all)
        acc = torch.zeros(_lib.NSCALARS, device=self.engine.device)
        preds, targs, feats = [], [], []
        for e, datas in self._iter_loaded(valid_loader, training=False):
            self._anchors(e, 2)
            e.forward(train=False, with_losses=True)
            acc += e.scalars
            preds.append(e.pred.clone())
            targs.append(e.labels.clone())
            if self.opt.save_best_features:
                feats.append([f.cpu() for f in e.feats.clone()])
        a = acc.cpu().numpy()
        nb = len(valid_loader)
        predictions = torch.cat(preds).cpu().numpy().reshape(-1, 1)
        targets = torch.cat(targs).cpu().numpy().reshape(-1, 1)
        score = self.get_score_from_result(predictions, targets)
        mis = [float(x) / nb for x in a[_lib.S2_MIS:_lib.S2_MIS + 8]]
        return float(a[_lib.S2_LOSS]) / nb, mis, score, predictions, targets, feats

    def compute_loss(self, outputs, labels, stage, C_F_all, F_F_all, T_F_all, A_F_all, V_F_all):
        """Solver.py:317-342 (MAE) -- losses evaluated on the features of the last forward."""
        predictions = outputs[0]
        task_loss = (predictions.reshape(-1) - torch.as_tensor(labels, device=predictions.device).reshape(-1)).abs().mean()
        return compute_custumized_loss(self.model, task_loss, outputs, labels, None, self.opt, stage, C_F_all, F_F_all,
                                       T_F_all, A_F_all, V_F_all)

    # ------------------------------------------------------------------ cold path
    def get_score_from_result(self, predictions, targets):
        """mae / corr of Utils.calc_metrics (Utils.py:135-136); the class-accuracy metrics are outside the hot path."""
        p, t = predictions.reshape(-1).astype(np.float64), targets.reshape(-1).astype(np.float64)
        mae = float(np.abs(p - t).mean())
        corr = float(np.corrcoef(p, t)[0, 1]) if p.std() > 0 and t.std() > 0 else 0.0
        return {"mae": mae, "corr": corr}

    def current_result_better(self, best, current):
        return best is None or current["mae"] < best["mae"]                                      # Solver.py:425-433

    # checkpoints: the reference's keys (Solver.py:57-62); 'model' = state_dict with the reference's parameter names,
    # the optimizers as flat Adam-moment buckets + device step counters (a complete, resumable optimizer state)
    def checkpoint(self, epoch):
        """Solver.py:57-62 ('epoch', 'model', 'optim_main', 'optim_vmi') + what a RESUMED run needs and the reference does not save:
        the five feature banks of the epoch (Solver.py:223-227 rebuilds them only by training: resuming with empty banks would apply
        the epoch-0 rule for a whole epoch), the device RNG step and numpy's global RNG state (host-drawn anchors,
        Model.py:81).  Optimizer format: 'm' / 'v' are dicts {reference parameter name: exp_avg / exp_avg_sq of that parameter's shape}
        (NOT a torch.optim.Adam state_dict: parameters are not separate tensors here), 'step' is Adam's state['step'], 'lr' the current
        rate; 'layout' is the fingerprint of the writer's flat-bucket layout (`HipEngine.layout_fingerprint`)."""
        st = self.engine.optimizer_state()
        nm = self.engine.named_moments()
        cpu = lambda d: {k: ({n: t.detach().cpu() for n, t in v.items()} if isinstance(v, dict) else v.detach().cpu()) for k, v in d.items()}
        n = int(self.engine.bank_rows)
        # 'm' / 'v' per parameter NAME (round 6, ADVICE r05: the flat buckets were stored raw and round 5 permuted the main bucket -- an
        # older checkpoint would have loaded without error onto the wrong parameters); 'layout' tags the bucket layout of the writer
        ck = {"epoch": epoch, "model": cpu(self.model.state_dict()), "layout": self.engine.layout_fingerprint(),
              "optim_main": cpu({"m": nm["main"]["m"], "v": nm["main"]["v"], "step": st["counters"][1:2], "lr": st["lr_main"]}),
              "optim_vmi": cpu({"m": nm["critic"]["m"], "v": nm["critic"]["v"], "step": st["counters"][2:3], "lr": st["lr_critic"]}),
              "rng_step": st["counters"][0:1].cpu(),
              "banks": {k: self.engine.bank[k][:n].detach().cpu().clone() for k in "CFTAV"} if n else None}
        if self._plateau is not None:
            ck["lr_plateau"] = [(s.lr, s.best, s.num_bad_epochs) for s in self._plateau]
        rs = np.random.get_state()
        ck["np_rng"] = (rs[0], torch.from_numpy(rs[1].astype(np.int64)), int(rs[2]), int(rs[3]), float(rs[4]))
        return ck

    def load_checkpoint(self, ck):
        """-> epoch of the checkpoint.  Restores parameters, both optimizers, the RNG counters and -- when the checkpoint has them --
        the feature banks (``self.resume_banks`` is then the tuple to pass to ``train(epoch + 1, loader, *banks)``)."""
        # weights_only=True: everything checkpoint() stores is a tensor / int / float / str / dict / tuple (the numpy RNG state as a
        # tuple of those) -- no pickle code execution on a checkpoint path
        ck = torch.load(ck, map_location="cpu", weights_only=True) if isinstance(ck, (str, os.PathLike)) else ck
        self.model.load_state_dict(ck["model"])
        cnt = torch.cat([ck["rng_step"].reshape(1), ck["optim_main"]["step"].reshape(1), ck["optim_vmi"]["step"].reshape(1),
                         torch.zeros(1, dtype=torch.int32)]).to(torch.int32)
        flat = {"counters": cnt, "lr_main": ck["optim_main"]["lr"], "lr_critic": ck["optim_vmi"]["lr"]}
        for grp, key, pre in (("main", "optim_main", "main"), ("critic", "optim_vmi", "crit")):
            m, v = ck[key]["m"], ck[key]["v"]
            if isinstance(m, dict):                     # per parameter name: layout-independent
                self.engine.load_named_moments(grp, m, v)
            elif ck.get("layout") == self.engine.layout_fingerprint():
                flat[pre + "_m"], flat[pre + "_v"] = m, v
            else:                                       # a raw flat bucket is only valid in the layout that wrote it
                raise _lib.MimrlError(f"checkpoint stores {key} as a flat bucket written under layout {ck.get('layout')!r}; this build's "
                                      f"layout is {self.engine.layout_fingerprint()!r} (the main bucket was re-ordered in round 5): "
                                      "refusing to assign Adam moments to the wrong parameters")
        self.engine.load_optimizer_state(flat)
        if self._plateau is not None and ck.get("lr_plateau") is not None:
            for s, (lr, best, bad) in zip(self._plateau, ck["lr_plateau"]):
                s.lr, s.best, s.num_bad_epochs = float(lr), float(best), int(bad)
        self.resume_banks = ([], [], [], [], [])
        b = ck.get("banks")
        if b is not None:
            self.engine.set_banks(*(b[k] for k in "CFTAV"))
            self.resume_banks = tuple(self.engine.bank[k][:self.engine.bank_rows] for k in "CFTAV")
        rs = ck.get("np_rng")
        if rs is not None:
            np.random.set_state((rs[0], rs[1].numpy().astype(np.uint32), rs[2], rs[3], rs[4]))
        return int(ck["epoch"])

    def save_results(self, best_predictions, best_targets, best_features, best_valid_state, best_test_state):
        """Solver.py:513-531."""
        os.makedirs(self.task_path, exist_ok=True)
        for name, arr in (("predictions_val", best_predictions[0]), ("predictions_test", best_predictions[1]),
                          ("predictions_test_for_valid", best_predictions[2]), ("targets_val", best_targets[0]),
                          ("targets_test", best_targets[1])):
            np.save(os.path.join(self.task_path, name + ".npy"), arr)
        if self.opt.save_best_features:
            import pickle
            for name, f in (("features_val", best_features[0]), ("features_test", best_features[1]),
                            ("features_test_for_valid", best_features[2])):
                with open(os.path.join(self.task_path, name + ".pkl"), "wb") as fh:
                    pickle.dump(f, fh)
        torch.save(best_valid_state, self.best_valid_model_path)
        torch.save(best_test_state, self.best_test_model_path)

    def solve(self):
        """Solver.py:38-105: epoch loop, best-valid / best-test tracking, checkpoints and prediction files (rank 0).
        TensorBoard and the class-accuracy metric tables are outside the hot path."""
        if self.rank == 0:
            os.makedirs(self.task_path, exist_ok=True)
            set_logger(os.path.join(self.task_path, "Running.log"))
            log_message(str(self.opt))
        banks = ([], [], [], [], [])
        best_score, best_predictions, best_features = [None, None, None], [None, None, None], [None, None, None]
        best_targets = [None, None]
        best_valid_state = best_test_state = None
        for epoch in range(self.opt.epochs_num):
            self.epoch = epoch
            r = self.train(epoch, self.train_loader, *banks)
            train_loss, train_loss_mi, train_mis, train_score = r[:4]
            banks = r[4:]
            val_loss, val_mis, val_score, val_pred, val_targ, val_feat = self.evaluate(self.valid_loader, *banks)
            test_loss, test_mis, test_score, test_pred, test_targ, test_feat = self.evaluate(self.test_loader, *banks)
            self.lr_schedule_step(val_loss)                                    # Solver.py:49-55
            if self.current_result_better(best_score[0], val_score):
                if self.rank == 0:
                    best_valid_state = self.checkpoint(epoch)
                best_score[0], best_predictions[0], best_features[0] = val_score, val_pred, val_feat
                best_score[2], best_predictions[2], best_features[2] = test_score, test_pred, test_feat
                best_targets[0] = val_targ
            if self.current_result_better(best_score[1], test_score):
                if self.rank == 0:
                    best_test_state = self.checkpoint(epoch)
                best_score[1], best_predictions[1], best_features[1] = test_score, test_pred, test_feat
                best_targets[1] = test_targ
            if self.rank == 0:
                log_message(f"Epoch:[{epoch + 1:3d}] || TrainLoss:[{train_loss:.3f}] TrainMILoss:[{train_loss_mi:.3f}] "
                            f"TrainMI_ft/fa/fv/in/st/sa/sv/cp:[" + "/".join(f"{m:.3f}" for m in train_mis) + "] "
                            f"Train_mae:[{train_score['mae']:6.3f}] || ValLoss:[{val_loss:.3f}] Val_mae:[{val_score['mae']:6.3f}] "
                            f"Val_corr:[{val_score['corr']:6.3f}] || TestLoss:[{test_loss:.3f}] Test_mae:[{test_score['mae']:6.3f}] "
                            f"Test_corr:[{test_score['corr']:6.3f}]")
        if self.rank == 0:
            log_message("Training complete.")
            for tag, sc in (("Best Valid Score", best_score[0]), ("Test Score at Best Valid", best_score[2]), ("Best Test Score", best_score[1])):
                log_message(tag + " " + " ".join(f"{k}:[{v:6.3f}]" for k, v in (sc or {}).items()))
            self.save_results(best_predictions, best_targets, best_features, best_valid_state, best_test_state)
        return best_score
