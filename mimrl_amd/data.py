"""Synthetic MOSI/MOSEI-shaped loader (the datasets, BERT weights and DataLoaderLocal are not shipped; SURVEY.md
section 0 item 8).  Batches use the reference's ``mosi_Dec`` 11-tuple layout (Customization.py:46, Solver.py:273-275)
with one substitution: slot 6 (``bert_sentences``) carries the precomputed BERT last-hidden-state features
[B,T,d_t] instead of token ids -- BERT itself is outside the hot path."""
import logging

import torch

from . import synth


class SyntheticLoader:
    """In-memory loader.  ``rank``/``world`` shard the samples by stride (rank, rank+world, ...) the way a
    DistributedSampler would, so that data-parallel replicas see disjoint data; a sharded loader always drops the ragged
    tail so that every rank runs the same number of (full) batches and collectives stay matched.  ``drop_last=False`` (the
    reference's default, Parameters.py:21) yields a shorter last batch."""

    def __init__(self, n, batch_size, T, d_t=768, d_a=74, d_v=35, seed=0, drop_last=False, device=None, ragged=False,
                 rank=0, world=1, pin=False):
        t, a, v, y = synth.synthetic_batch(n, T, d_t, d_a, d_v, seed=seed, ragged=ragged)
        if world > 1:
            per = n // world
            t, a, v, y = (x[rank::world][:per] for x in (t, a, v, y))
            drop_last = True
        mk = lambda x: torch.from_numpy(x.copy()).to(device) if device else (torch.from_numpy(x.copy()).pin_memory() if pin else torch.from_numpy(x.copy()))
        self.t, self.a, self.v, self.y = mk(t), mk(a), mk(v), mk(y)
        self.n, self.bs, self.drop_last, self.T = int(self.y.shape[0]), batch_size, drop_last, T

    def __len__(self):
        return self.n // self.bs if self.drop_last else (self.n + self.bs - 1) // self.bs

    def num_samples(self):
        return len(self) * self.bs if self.drop_last else self.n

    def __iter__(self):
        for i in range(len(self)):
            s = slice(i * self.bs, min((i + 1) * self.bs, self.n))
            B = s.stop - s.start
            ones = torch.ones(B, self.T, dtype=torch.long)
            yield (None, self.a[s], self.v[s], None, None, self.y[s].reshape(-1, 1), self.t[s], torch.zeros_like(ones), ones,
                   None, None)


def get_data_loader(opt, rank=0, world=1):
    """-> (train, valid, test, d_t, d_a, d_v) like DataLoaderUniversal.get_data_loader (DataLoaderUniversal.py:10-95).
    Only synthetic data exists here: ``--dataset synthetic`` is the honest name; the reference's ``mosi_Dec`` / ``mosei_Dec``
    names are accepted (they select the 'Dec' batch layout) but produce SYNTHETIC data too, and say so loudly.
    The training set is sharded over data-parallel ranks; valid / test are replicated (every rank evaluates all of it)."""
    if opt.dataset not in ("synthetic", "mosi_Dec", "mosei_Dec"):
        raise NotImplementedError(f"--dataset {opt.dataset}: only synthetic MOSI/MOSEI-shaped data is available "
                                  f"(the reference's pickles and DataLoaderLocal are not shipped)")
    if opt.dataset != "synthetic":
        logging.warning("--dataset %s: the %s pickles are not shipped; training on SYNTHETIC %s-shaped random triples "
                        "(scores are NOT dataset results). Use --dataset synthetic to silence this.", opt.dataset,
                        opt.dataset, opt.dataset.split("_")[0])
    n = int(getattr(opt, "synthetic_n", 1284))
    d_t, d_a, d_v = int(getattr(opt, "d_t", 768)), int(getattr(opt, "d_a", 74)), int(getattr(opt, "d_v", 35))
    drop = bool(getattr(opt, "drop_last", False))
    pin = torch.cuda.is_available()
    # 288 GB of HBM: MOSI-sized features are 0.2 GB, MOSEI-sized 2.9 GB -- the whole dataset lives on the device and a batch is a
    # device-to-device slice copy on the upload stream (no PCIe on the step path) unless --host_data or it would not fit
    device = None
    if pin and not getattr(opt, "host_data", False):
        need = 1.5 * n * opt.time_len * (d_t + d_a + d_v) * 4
        if need < 0.25 * torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory:
            device = torch.device("cuda", torch.cuda.current_device())
    k = int(getattr(opt, "k_neighbor", 2))

    def mk(m, seed, r, w):
        if 0 < m % opt.batch_size < k:      # a last batch with fewer than k_neighbor rows has no kNN product sample (the
            m -= m % opt.batch_size          # reference's sklearn call raises on it, Model.py:85-86): do not generate one
        return SyntheticLoader(m, opt.batch_size, opt.time_len, d_t, d_a, d_v, seed=seed, drop_last=drop, rank=r, world=w, pin=pin,
                               device=device)

    return (mk(n, opt.seed, rank, world), mk(max(n // 6, opt.batch_size), opt.seed + 1, 0, 1),
            mk(max(n // 3, opt.batch_size), opt.seed + 2, 0, 1), d_t, d_a, d_v)
