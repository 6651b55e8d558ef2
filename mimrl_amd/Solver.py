"""``Solver`` with the reference's surface (Solver.py:18-531) re-expressed around two fused device steps.

New (named by BASELINE.json's north star, absent in the reference): ``Solver.step(datas)`` = one stage-1 update
(body of Solver.py:205-214) + one stage-2 update (body of :221-236) on one batch.  ``train()`` keeps the reference's
epoch ordering -- ``stage1_n`` full critic passes (skipped in epoch 0), then one model pass that also builds the next
epoch's feature banks -- but accumulates losses / MI terms / banks on the device and reads back once per epoch
(the reference performs >= 10 ``.item()`` syncs and 18 bank D2H copies per iteration).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib, dist as mdist, synth
from .Customization import compute_custumized_loss, compute_outputs_from_model, other_model_operations
from .Model import Model
from .Utils import log_message, set_logger


class Solver:
    def __init__(self, opt, loaders=None):
        self.opt = opt
        self.world, self.rank, self.local_rank = mdist.init_from_env()
        if opt.loss != "MAE":
            raise NotImplementedError(f"--loss {opt.loss}: only MAE is on the MI355X hot path (SURVEY.md section 2)")
        if opt.optm != "Adam":
            raise NotImplementedError(f"--optm {opt.optm}: only Adam is fused on the MI355X hot path")
        if loaders is None:
            from .data import get_data_loader
            loaders = get_data_loader(opt)
        self.train_loader, self.valid_loader, self.test_loader, self.d_t, self.d_a, self.d_v = loaders
        if torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)
        cap = len(self.train_loader) * opt.batch_size * self.world
        self.model = Model(opt, self.d_t, self.d_a, self.d_v, bank_capacity=cap)
        other_model_operations(self.model, opt)
        self.engine = self.model.engine
        if self.world > 1:      # identical replicas: rank 0's initial parameters everywhere
            mdist.broadcast_(self.engine.main["p"])
            mdist.broadcast_(self.engine.crit["p"])
            self.engine.params_changed()
        self.base_lr = float(opt.learning_rate)
        self.epoch = 0

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
        raise NotImplementedError(f"--lr_decrease {o.lr_decrease}")

    def _apply_lr(self, epoch: int):
        f = self.lr_factor(epoch)
        self.engine.set_lr(self.base_lr * f, self.base_lr * float(self.opt.mi_lr_rate) * f)

    # ------------------------------------------------------------------ the hot path
    def get_label_from_datas(self, datas):
        return datas[5]                                                       # Solver.py:273-275 ('Dec' layout)

    def _load(self, datas):
        _, a, v, _, _, labels, feats, _, _, _, _ = datas
        self.engine.set_batch(feats, a, v, labels)

    def _anchors(self, stage):
        e = self.engine
        if e.bank_rows > 0 and not e.cfg.device_anchors:
            e.set_anchors(stage, synth.draw_anchors(e.bank_rows, e.m_anchor, 6))

    def stage1_step(self, datas=None, draw_anchors=True):
        """Critic update (Solver.py:205-214).  Returns the stage-1 loss as a device scalar."""
        if datas is not None:
            self._load(datas)
        if draw_anchors:
            self._anchors(1)
        if self.world > 1:
            mdist.ddp_stage_step(self.engine, 1, self.world)
        else:
            self.engine.stage1_step()
        return self.engine.scalars[_lib.S1_LOSS]

    def stage2_step(self, datas=None, draw_anchors=True):
        """Model update (Solver.py:221-236).  Returns (loss, mis[8], pred[B,1]) as device tensors."""
        if datas is not None:
            self._load(datas)
        if draw_anchors:
            self._anchors(2)
        if self.world > 1:
            mdist.ddp_stage_step(self.engine, 2, self.world)
        else:
            self.engine.stage2_step()
        s = self.engine.scalars
        return s[_lib.S2_LOSS], s[_lib.S2_MIS:_lib.S2_MIS + 8], self.engine.pred.reshape(-1, 1)

    def _prefetch(self, on):
        if getattr(self, "_prefetch_on", False) != on:
            self.engine.set_stage2_prefetch(on)
            self._prefetch_on = on

    def step(self, datas):
        """One two-stage iteration on one batch: the unit BASELINE.json's metric counts.  Both stages see the same
        batch and stage 1 leaves the main model untouched, so the engine runs the stage-2 forward pass beside stage 1
        (`mimrl_set_stage2_prefetch`); the numbers are those of the sequential order."""
        self._prefetch(True)
        self._load(datas)
        self._anchors(1)          # host-drawn anchors (if any) for BOTH stages go up before stage 1: in overlap mode the
        self._anchors(2)          # stage-2 kNN sampler already runs beside stage 1 (same draw order as the reference)
        if self.world > 1:
            l1 = self.stage1_step(draw_anchors=False)
            l2, mis, pred = self.stage2_step(draw_anchors=False)
            return l1, l2, mis, pred
        self.engine.step()        # mimrl_two_stage_step: both stages as one captured graph
        sc = self.engine.scalars
        return sc[_lib.S1_LOSS], sc[_lib.S2_LOSS], sc[_lib.S2_MIS:_lib.S2_MIS + 8], self.engine.pred.reshape(-1, 1)

    # ------------------------------------------------------------------ Solver.train (Solver.py:194-248)
    def train(self, epoch, train_loader, C_F_all, F_F_all, T_F_all, A_F_all, V_F_all):
        self.model.train()
        e = self.engine
        self._prefetch(False)                                                  # epoch-ordered passes: stages see different batches
        self._apply_lr(epoch)
        e.set_banks(C_F_all, F_F_all, T_F_all, A_F_all, V_F_all)
        dev = e.device
        acc = torch.zeros(_lib.NSCALARS, device=dev)
        nb = len(train_loader)
        if epoch > 0 and len(C_F_all) > 0:                                     # Solver.py:200-203: epoch 0 skips stage 1
            for _ in range(self.opt.stage1_n):
                for datas in self.train_loader:
                    self.stage1_step(datas)
                    acc[_lib.S1_LOSS] += e.scalars[_lib.S1_LOSS]
        B = self.opt.batch_size
        newC = torch.empty(nb * B, 1, device=dev)
        newF, newT, newA, newV = (torch.empty(nb * B, 128, device=dev) for _ in range(4))
        preds = torch.empty(nb * B, device=dev)
        targs = torch.empty(nb * B, device=dev)
        for i, datas in enumerate(train_loader):
            self.stage2_step(datas)
            sl = slice(i * B, (i + 1) * B)
            newC[sl, 0] = e.labels                                            # Solver.py:223-227 (features of THIS pass)
            newF[sl], newT[sl], newA[sl], newV[sl] = e.feats[0], e.feats[1], e.feats[2], e.feats[3]
            preds[sl], targs[sl] = e.pred, e.labels
            acc[32:] += e.scalars[32:]
        if self.world > 1:                                                    # banks are replicated
            newC, newF, newT, newA, newV = (mdist.allgather_rows(x, self.world) for x in (newC, newF, newT, newA, newV))
        a = acc.cpu().numpy()                                                 # the ONE read-back of the epoch
        predictions, targets = preds.cpu().numpy().reshape(-1, 1), targs.cpu().numpy().reshape(-1, 1)
        train_score = self.get_score_from_result(predictions, targets)
        mis = [float(x) / nb for x in a[_lib.S2_MIS:_lib.S2_MIS + 8]]
        return (float(a[_lib.S2_LOSS]) / nb, float(a[_lib.S1_LOSS]) / nb, mis, train_score, newC, newF, newT, newA, newV)

    # ------------------------------------------------------------------ Solver.evaluate (Solver.py:250-270)
    def evaluate(self, valid_loader, C_F_all, F_F_all, T_F_all, A_F_all, V_F_all):
        self.model.eval()
        e = self.engine
        e.set_banks(C_F_all, F_F_all, T_F_all, A_F_all, V_F_all)
        acc = torch.zeros(_lib.NSCALARS, device=e.device)
        preds, targs, feats = [], [], []
        for datas in valid_loader:
            self._load(datas)
            self._anchors(2)
            e.forward(train=False, with_losses=True)
            acc += e.scalars
            preds.append(e.pred.clone())
            targs.append(e.labels.clone())
            if self.opt.save_best_features:
                feats.append(e.feats.clone())
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
        p, t = predictions.reshape(-1).astype(np.float64), targets.reshape(-1).astype(np.float64)
        mae = float(np.abs(p - t).mean())
        corr = float(np.corrcoef(p, t)[0, 1]) if p.std() > 0 and t.std() > 0 else 0.0
        return {"mae": mae, "corr": corr}

    def current_result_better(self, best, current):
        return best is None or current["mae"] < best["mae"]

    def solve(self):
        """Solver.py:38-105 (epoch loop; logging/saving reduced to the essentials)."""
        if self.rank == 0:
            task_path = os.path.join("./TaskRuning", self.opt.task_name)
            os.makedirs(task_path, exist_ok=True)
            set_logger(os.path.join(task_path, "Running.log"))
            log_message(str(self.opt))
        banks = ([], [], [], [], [])
        best = [None, None]
        for epoch in range(self.opt.epochs_num):
            r = self.train(epoch, self.train_loader, *banks)
            train_loss, train_loss_mi, train_mis, train_score = r[:4]
            banks = r[4:]
            val_loss, val_mis, val_score, *_ = self.evaluate(self.valid_loader, *banks)
            test_loss, test_mis, test_score, *_ = self.evaluate(self.test_loader, *banks)
            if self.current_result_better(best[0], val_score):
                best[0], best[1] = val_score, test_score
            if self.rank == 0:
                log_message(f"Epoch {epoch:3d} | train loss {train_loss:.4f} mi-loss {train_loss_mi:.4f} mae {train_score['mae']:.4f} "
                            f"| valid loss {val_loss:.4f} mae {val_score['mae']:.4f} | test loss {test_loss:.4f} mae {test_score['mae']:.4f} "
                            f"| mi ft/fa/fv/in/st/sa/sv/cp " + "/".join(f"{m:.4f}" for m in train_mis))
        if self.rank == 0:
            log_message(f"Training complete. best valid {best[0]} / test at best valid {best[1]}")
        return best
