"""Flag parsers and tiny helpers of the hot path (reference: Utils.py:226-248, 297-298, 307-311)."""
import argparse
import logging
import os

import torch


def str2bool(v: str) -> bool:
    t = v.strip().lower()
    if t in {"yes", "true", "t", "y", "1"}:
        return True
    if t in {"no", "false", "f", "n", "0"}:
        return False
    raise argparse.ArgumentTypeError("Boolean value expected." + v)


def str2bools(v: str):
    """'1-0-1' -> [True, False, True]"""
    return [str2bool(x) for x in v.split("-")]


def str2floats(v: str):
    """'0.1-0.2' -> [0.1, 0.2]"""
    return [float(x) for x in v.split("-")]


def str2listoffints(v: str):
    """'50-3-128=10-3-128' -> [[50,3,128],[10,3,128]]"""
    return [[int(x) for x in blk.split("-")] for blk in v.split("=")]


def get_mask_from_sequence(sequence: torch.Tensor, dim: int) -> torch.Tensor:
    """True where a row is all-zero (padding); Utils.py:297-298."""
    return sequence.abs().sum(dim=dim) == 0


def to_gpu(x, on_cpu: bool = False, gpu_id=None):
    """Utils.py:307-311."""
    if torch.cuda.is_available() and not on_cpu:
        x = x.cuda(gpu_id)
    return x


def set_logger(log_path: str):
    logger = logging.getLogger()
    logger.setLevel(logging.DEBUG)
    if not logger.handlers:
        os.makedirs(os.path.dirname(log_path) or ".", exist_ok=True)
        fh = logging.FileHandler(log_path)
        fh.setFormatter(logging.Formatter("%(asctime)s:%(levelname)s: %(message)s"))
        sh = logging.StreamHandler()
        sh.setFormatter(logging.Formatter("%(message)s"))
        logger.addHandler(fh)
        logger.addHandler(sh)


def log_message(message: str):
    logging.log(msg=message, level=logging.DEBUG)
