"""Diagnostic: the block-0 K-axis LayerNorm gain gradient (3 values) of the cfg2 stage-2 pass over several fresh engines."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tests.test_gpu_step import _bench_engine   # noqa: E402

anchors = None
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    opt, N, batch, banks, eng = _bench_engine("cfg2", "bf16", False, device_anchors=False)
    if anchors is None:
        rng = np.random.default_rng(5)
        anchors = np.stack([rng.choice(N, size=opt.batch_size // opt.k_neighbor, replace=False) for _ in range(6)])
    eng.set_anchors(2, anchors)
    eng.stage_grads(2)
    torch.cuda.synchronize()
    g = eng.grads
    print(r, g["mlp_encoder.layers_stack.0.ln_k.weight"].cpu().numpy(), g["mlp_encoder.layers_stack.0.ln_k.bias"].cpu().numpy(),
          g["mlp_encoder.layers_stack.0.mlp_k.fc1.bias"].cpu().numpy())
    eng.close()
