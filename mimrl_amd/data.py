"""Synthetic MOSI/MOSEI-shaped loader (the datasets, BERT weights and DataLoaderLocal are not shipped; SURVEY.md
section 0 item 8).  Batches use the reference's ``mosi_Dec`` 11-tuple layout (Customization.py:46, Solver.py:273-275)
with one substitution: slot 6 (``bert_sentences``) carries the precomputed BERT last-hidden-state features
[B,T,d_t] instead of token ids -- BERT itself is outside the hot path."""
import numpy as np
import torch

from . import synth


class SyntheticLoader:
    def __init__(self, n, batch_size, T, d_t=768, d_a=74, d_v=35, seed=0, drop_last=True, device=None, ragged=False):
        t, a, v, y = synth.synthetic_batch(n, T, d_t, d_a, d_v, seed=seed, ragged=ragged)
        dev = device
        self.t, self.a, self.v = (torch.from_numpy(x).to(dev) if dev else torch.from_numpy(x) for x in (t, a, v))
        self.y = torch.from_numpy(y).to(dev) if dev else torch.from_numpy(y)
        self.n, self.bs, self.drop_last = n, batch_size, drop_last
        self.T = T

    def __len__(self):
        return self.n // self.bs if self.drop_last else (self.n + self.bs - 1) // self.bs

    def __iter__(self):
        for i in range(len(self)):
            s = slice(i * self.bs, min((i + 1) * self.bs, self.n))
            B = s.stop - s.start
            ones = torch.ones(B, self.T, dtype=torch.long)
            yield (None, self.a[s], self.v[s], None, None, self.y[s].reshape(-1, 1), self.t[s], torch.zeros_like(ones), ones,
                   None, None)


def get_data_loader(opt):
    """-> (train, valid, test, d_t, d_a, d_v) like DataLoaderUniversal.get_data_loader (DataLoaderUniversal.py:10-95);
    only ``--dataset synthetic`` exists here."""
    if opt.dataset not in ("synthetic", "mosi_Dec", "mosei_Dec"):
        raise NotImplementedError(f"--dataset {opt.dataset}: only synthetic MOSI/MOSEI-shaped data is available "
                                  f"(the reference's pickles and DataLoaderLocal are not shipped)")
    n = int(getattr(opt, "synthetic_n", 1284))
    d_t, d_a, d_v = int(getattr(opt, "d_t", 768)), int(getattr(opt, "d_a", 74)), int(getattr(opt, "d_v", 35))
    mk = lambda m, seed: SyntheticLoader(m, opt.batch_size, opt.time_len, d_t, d_a, d_v, seed=seed, drop_last=True)
    return mk(n, opt.seed), mk(max(n // 6, opt.batch_size), opt.seed + 1), mk(max(n // 3, opt.batch_size), opt.seed + 2), d_t, d_a, d_v
