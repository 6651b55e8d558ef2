"""Fixture configurations shared by make_golden.py (reference side) and the parity tests."""
from types import SimpleNamespace

CONFIGS = {
    # small, fast: every test tier can afford it
    "tiny_sep": dict(B=8, T=6, N=40, seed=1, critic="separate", cube="6-3-128=4-3-128", traj=3),
    "tiny_cat": dict(B=8, T=6, N=40, seed=2, critic="concat", cube="6-3-128=4-3-128", traj=2),
    # ragged a/v lengths (zero rows) -> packed-sequence semantics of Model.py:425-447
    "tiny_ragged": dict(B=8, T=6, N=40, seed=3, critic="separate", cube="6-3-128=4-3-128", traj=1, ragged=True),
    # ln_first CubeMLP variant + hardtanh CMI head + nwj bound (flag-reachable alternatives)
    "tiny_alt": dict(B=8, T=6, N=40, seed=4, critic="concat", cube="6-3-128=4-3-128", traj=1,
                     ln_first=True, cmi_last="hardtanh", bound="nwj"),
    # --encoders conv (Conv1d k=3 over time instead of the bi-GRUs, Model.py:247-249,437-439), ragged inputs
    "tiny_conv": dict(B=8, T=6, N=40, seed=5, critic="separate", cube="6-3-128=4-3-128", traj=2, ragged=True, encoders="conv"),
    # mine bound: its loss term is not -mi (Model.py:121-125), and stage 2 mixes both forms (Model.py:386)
    "tiny_mine": dict(B=8, T=6, N=40, seed=6, critic="separate", cube="6-3-128=4-3-128", traj=2, bound="mine"),
    # baselines (VMI.py:72-110) of the two bounds that read them: trainable (unnormalized) and Gaussian
    "tiny_tuba_un": dict(B=8, T=6, N=40, seed=10, critic="separate", cube="6-3-128=4-3-128", traj=2, bound="tuba",
                         baseline="unnormalized"),
    "tiny_interp_ga": dict(B=8, T=6, N=40, seed=11, critic="separate", cube="6-3-128=4-3-128", traj=2, bound="interpolate",
                           baseline="gaussain"),
    # --encoders lstm (1-layer bi-LSTM, Model.py:250-252), ragged inputs
    "tiny_lstm": dict(B=8, T=6, N=40, seed=9, critic="separate", cube="6-3-128=4-3-128", traj=2, ragged=True, encoders="lstm"),
    # --encoders lstm at BASELINE cfg1's shape (B = 32, T = 50, ragged lengths: four rows of different lengths per recurrence workgroup):
    # the MFMA LSTM kernels of round 5 (lstm.hip) against the reference's nn.LSTM on packed sequences
    "cfg1_lstm": dict(B=32, T=50, N=1000, seed=17, critic="separate", cube="50-3-128=10-3-128", traj=2, ragged=True, encoders="lstm"),
    # interpolated bound (VMI.py:201-250) with the constant baseline, concat critic
    "tiny_interp": dict(B=8, T=6, N=40, seed=8, critic="concat", cube="6-3-128=4-3-128", traj=2, bound="interpolate"),
    # awkward sizes: batch not a multiple of the tile sizes, inputs shorter than --time_len (zero padding of the cube,
    # Model.py:468-470), k_neighbor=3, ragged lengths
    "tiny_odd": dict(B=12, T=5, L=8, N=50, seed=7, critic="separate", cube="8-3-128=3-3-128", traj=2, ragged=True, k=3),
    # BASELINE cfg1: B=32, T=50, canonical README flags, N=1000 as in the reference smoke test (Model.py:607)
    "cfg1_sep": dict(B=32, T=50, N=1000, seed=0, critic="separate", cube="50-3-128=10-3-128", traj=6),
    "cfg1_cat": dict(B=32, T=50, N=1000, seed=0, critic="concat", cube="50-3-128=10-3-128", traj=1),
    # N1 at benchmark scale: cfg1 / cfg2 with RAGGED a/v lengths -- packed-sequence GRU semantics (Model.py:425-447) with
    # four batch rows of different lengths per recurrence workgroup at T = 50
    "cfg1_ragged": dict(B=32, T=50, N=1000, seed=15, critic="separate", cube="50-3-128=10-3-128", traj=2, ragged=True),
    "cfg2_ragged": dict(B=128, T=50, N=1284, seed=16, critic="separate", cube="50-3-128=10-3-128", traj=1, ragged=True),
    # BASELINE cfg2 at FULL size (the bench configuration: B=128, T=50, MOSI-sized banks)
    "cfg2_sep": dict(B=128, T=50, N=1284, seed=0, critic="separate", cube="50-3-128=10-3-128", traj=1),
    # BASELINE cfg3 reduced in the batch only: T = time_len = 500 (L-axis MLP 500 -> 50), concat critic, k=2
    "cfg3_small": dict(B=16, T=500, N=400, seed=0, critic="concat", cube="50-3-128=10-3-128", traj=1),
    # BASELINE cfg5, reference-supported subset (gru, d_common=128, fp32), T = 1000: 2000 serial cell steps per pass
    "cfg5_small": dict(B=8, T=1000, N=163, seed=0, critic="separate", cube="50-3-128=10-3-128", traj=1),
    # BASELINE cfg3 at FULL size (B=256, T=500, concat critic, MOSEI-sized banks N=16326) and the cfg5 subset at the size the step tests
    # run (B=32, T=1000): one reference step each, run ONCE in the build container (minutes of CPU autograd).  `slices`: per-tensor
    # gradient norms / sums for every tensor plus a 512-entry strided slice of each (the fixture stays small); no forward feature dump.
    "cfg3_full": dict(B=256, T=500, N=16326, seed=0, critic="concat", cube="50-3-128=10-3-128", traj=1, slices=True),
    "cfg5_full": dict(B=32, T=1000, N=163, seed=0, critic="separate", cube="50-3-128=10-3-128", traj=1, slices=True),
    # --features_compose_t/k sum (Model.py:473-485)
    "tiny_sum": dict(B=8, T=6, N=40, seed=12, critic="separate", cube="6-3-128=4-3-128", traj=1, compose="sum"),
    # DISCRETE labels (steps of 0.2 like MOSI's annotator averages) in the batch and in the label bank: the R^1 kNN of the
    # ta_c / tv_c estimators is then decided by scikit-learn's KDTree tie order (Model.py:82-86)
    "tiny_disc": dict(B=8, T=6, N=120, seed=13, critic="separate", cube="6-3-128=4-3-128", traj=2, discrete=True),
    "cfg1_disc": dict(B=32, T=50, N=1000, seed=14, critic="separate", cube="50-3-128=10-3-128", traj=1, discrete=True),
}

# Epoch-level fixtures: the reference's own Solver.train / Solver.evaluate (Solver.py:194-270) over several epochs.
# n_*: dataset sizes (a size that is not a multiple of B exercises the partial last batch: drop_last defaults to False,
# Parameters.py:21); stage1_n critic passes per epoch; multi-step lr schedule stepping at the given epochs.
EPOCH_CONFIGS = {
    "epoch_tiny": dict(B=8, T=6, seed=21, critic="separate", cube="6-3-128=4-3-128", n_train=32, n_valid=16, n_test=8,
                       epochs=3, stage1_n=2, lr=1e-4, lr_iter="1-2", lr_rate=0.5),
    "epoch_tail": dict(B=8, T=6, seed=22, critic="separate", cube="6-3-128=4-3-128", n_train=36, n_valid=12, n_test=8,
                       epochs=2, stage1_n=1, lr=1e-4, lr_iter="1-2", lr_rate=0.5),
}


def grad_slice_index(numel, n=512):
    """Entries of a flattened gradient tensor a `slices` fixture stores: all of it up to n entries, else n evenly strided ones."""
    import numpy as np
    return np.arange(numel) if numel <= n else (np.arange(n, dtype=np.int64) * numel) // n


def parse_cube(s):
    return [list(map(int, blk.split("-"))) for blk in s.split("=")]


def make_opt(c):
    """The subset of Parameters.py flags the hot path reads, README values."""
    cube = parse_cube(c["cube"])
    return SimpleNamespace(
        batch_size=c["B"], d_common=128, encoders=c.get("encoders", "gru"), features_compose_t=c.get("compose", "mean"),
        features_compose_k=c.get("compose", "mean"),
        num_class=1, activate="gelu", time_len=c.get("L", c["T"]), d_hiddens=cube, d_outs=cube,
        dropout_mlp=[0.0, 0.0, 0.0], dropout=[0.0, 0.0, 0.0, 0.0], bias=True, ln_first=c.get("ln_first", False),
        res_project=[True] * len(cube), critic_type=c["critic"], baseline_type=c.get("baseline", "constant"),
        bound_type=c.get("bound", "infonce"), loss_mi_coefficient1=[1.0] * 11, loss_mi_coefficient2=[0.01] * 8,
        mi_lr_rate=1.0, cmi_lr_rate=1.0, k_neighbor=c.get("k", 2), radius=1.0, cmi_last_acticate=c.get("cmi_last", "sigmoid"),
        stage1_n=c.get("stage1_n", 1), loss="MAE", gradient_clip=1.5, optm="Adam", learning_rate=c.get("lr", 4e-3), weight_decay=0.0,
        lr_decrease="multi_step", lr_decrease_iter=c.get("lr_iter", "50-60"), lr_decrease_rate=c.get("lr_rate", 0.1),
        dataset="mosi_Dec", parallel=True, text="none",
    )


def epoch_data(c):
    """Synthetic train / valid / test sets of an epoch fixture -> dict of (t, a, v, y) numpy tuples."""
    from mimrl_amd import synth
    return {k: synth.synthetic_batch(c["n_" + k], c["T"], seed=c["seed"] + i) for i, k in enumerate(("train", "valid", "test"))}


def split_batches(data, B):
    """[(t, a, v, y)] in loader order, last batch partial (drop_last=False, Parameters.py:21)."""
    n = data[3].shape[0]
    return [tuple(x[i:i + B] for x in data) for i in range(0, n, B)]


def quantize_labels(c, batch, banks):
    """``discrete`` fixtures: labels and the label bank rounded to multiples of 0.2 (in place on copies)."""
    import numpy as np
    if not c.get("discrete"):
        return batch, banks
    q = lambda x: (np.round(np.asarray(x, np.float32) * 5) / 5).astype(np.float32)
    t, a, v, y = batch
    banks = dict(banks)
    banks["C"] = q(banks["C"])
    return (t, a, v, q(y)), banks
