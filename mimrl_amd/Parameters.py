"""Command line of the runner: the SAME flag names, types and defaults as the reference's Parameters.py:8-70
(including its spelling ``--cmi_last_acticate``), table-driven.  Extra MI355X flags are listed last."""
import argparse

from .Utils import str2bools, str2floats, str2listoffints

_STR, _INT, _FLT, _FLAG = "s", "i", "f", "flag"
# (name, kind-or-type, default[, choices])
_REFERENCE_FLAGS = [
    ("task_name", _STR, "test"),
    ("dataset", _STR, "mosi_SDK"), ("normalize", str2bools, "0-0-0"), ("log_scale", str2bools, "0-0-0"),
    ("text", _STR, "text"), ("audio", _STR, "covarep"), ("video", _STR, "facet41"),
    ("batch_size", _INT, 16), ("num_workers", _INT, 4), ("persistent_workers", _FLAG, None), ("pin_memory", _FLAG, None),
    ("drop_last", _FLAG, None), ("task", _STR, "regression", ["classification", "regression"]), ("num_class", _INT, 1),
    ("d_common", _INT, 128), ("encoders", _STR, "gru"), ("features_compose_t", _STR, "mean"),
    ("features_compose_k", _STR, "mean"), ("activate", _STR, "gelu"), ("time_len", _INT, 100),
    ("d_hiddens", str2listoffints, "10-2-128=5-2-128"), ("d_outs", str2listoffints, "10-2-128=5-2-128"),
    ("dropout_mlp", str2floats, "0.5-0.5-0.5"), ("dropout", str2floats, "0.5-0.5-0.5-0.5"), ("bias", _FLAG, None),
    ("ln_first", _FLAG, None), ("res_project", str2bools, "1-1"),
    ("critic_type", _STR, "separate"), ("baseline_type", _STR, "constant"), ("bound_type", _STR, "infonce"),
    ("loss_mi_coefficient1", str2floats, "-".join(["0.1"] * 11)), ("loss_mi_coefficient2", str2floats, "-".join(["0.1"] * 8)),
    ("mi_lr_rate", _FLT, 1.0), ("cmi_lr_rate", _FLT, 1.0), ("k_neighbor", _INT, 2), ("radius", _FLT, 1.0),
    ("cmi_last_acticate", _STR, "sigmoid", ["hardtanh", "sigmoid"]), ("stage1_n", _INT, 1),
    ("seed", _INT, 0), ("loss", _STR, "MAE", ["Focal", "CE", "BCE", "RMSE", "MSE", "SIMSE", "MAE", "CCC"]),
    ("gradient_clip", _FLT, 1.0), ("epochs_num", _INT, 2), ("optm", _STR, "Adam", ["SGD", "SAM", "Adam"]),
    ("learning_rate", _FLT, 4e-3), ("bert_freeze", _STR, "no", ["part", "no", "all"]), ("bert_lr_rate", _FLT, -1),
    ("weight_decay", _FLT, 0.0), ("lr_decrease", _STR, "step", ["multi_step", "step", "exp", "plateau"]),
    ("lr_decrease_iter", _STR, "60"), ("lr_decrease_rate", _FLT, 0.1), ("save_best_features", _FLAG, None),
    ("print_params", _FLAG, None), ("check_gradient", _FLAG, None), ("parallel", _FLAG, None), ("cuda", _STR, "0"),
]
# MI355X-side additions (not in the reference)
_EXTRA_FLAGS = [
    ("precision", _STR, "fp32", ["fp32", "bf16"]),      # MFMA operand type (accumulation/state/optimizer always fp32)
    ("no_graph", _FLAG, None),                          # replay the two stages as hipGraphs unless set
    ("host_anchors", _FLAG, None),                      # draw kNN anchors with numpy's global RNG exactly like Model.py:81
    ("host_data", _FLAG, None),                         # keep the dataset in (pinned) host memory; default: resident in HBM when it fits
    ("synthetic_n", _INT, 1284),                        # --dataset synthetic: number of training samples (MOSI-sized)
    ("d_t", _INT, 768), ("d_a", _INT, 74), ("d_v", _INT, 35),
]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="MIMRL two-stage training on MI355X (reference-compatible flags)")
    for spec in _REFERENCE_FLAGS + _EXTRA_FLAGS:
        name, kind, default = spec[0], spec[1], spec[2]
        kw = {}
        if len(spec) > 3:
            kw["choices"] = spec[3]
        if kind == _FLAG:
            p.add_argument("--" + name, action="store_true")
            continue
        kw["type"] = {_STR: str, _INT: int, _FLT: float}.get(kind, kind)
        kw["default"] = default
        p.add_argument("--" + name, **kw)
    return p


def parse_args(argv=None):
    """Same contract as the reference's parse_args(): list-typed defaults are parsed through their type."""
    return build_parser().parse_args(argv)


if __name__ == "__main__":
    print(parse_args())
