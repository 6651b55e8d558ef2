"""Glue with the reference's function names (Customization.py:28-115)."""
import torch

from . import _lib


def other_model_operations(model, opt):
    """Customization.py:28-37: BERT freezing / tokenizer are outside the hot path; orthogonal ``weight_hh`` init is
    part of Model's default initialiser (synth.default_tensor)."""
    return None


def compute_outputs_from_model(model, datas, opt):
    """Customization.py:44-51 ('Dec' branch).  datas[6] carries the BERT features [B,T,d_t] (see data.py)."""
    if "Dec" not in opt.dataset and opt.dataset != "synthetic":
        raise NotImplementedError(f"--dataset {opt.dataset} is not on the MI355X hot path")
    _, a_data, v_data, _, _, labels, bert_feats, bert_types, bert_mask, _, _ = datas
    return model(bert_feats, bert_types, bert_mask, a_data, v_data, return_features=True, labels=labels)


def compute_custumized_loss(model, task_loss, outputs, labels, loss_functions, opt, stage, C_F_all, F_F_all, T_F_all,
                            A_F_all, V_F_all):
    """Customization.py:91-115: stage switch, empty-bank rule, weighted sum.  Returns device scalars (no host sync)."""
    predictions, F_F, T_F, A_F, V_F = outputs
    zero = lambda: torch.zeros((), device=predictions.device)
    if stage not in (1, 2):
        raise NotImplementedError
    if len(C_F_all) == 0:                                                        # Customization.py:97-98,105-106
        return (zero() if stage == 1 else task_loss), [zero() for _ in range(8)]
    m = model.module
    if stage == 1:
        mis, mi_losses = m.compute_vmi_loss_stage1(predictions, labels, F_F, T_F, A_F, V_F, C_F_all, F_F_all, T_F_all, A_F_all, V_F_all)
        return m.engine.scalars[_lib.S1_LOSS], mis
    mis, mi_losses = m.compute_vmi_loss_stage2(predictions, labels, F_F, T_F, A_F, V_F, C_F_all, F_F_all, T_F_all, A_F_all, V_F_all)
    return m.engine.scalars[_lib.S2_LOSS], mis
