"""Shared test helpers: golden fixtures, configs, error metrics."""
import ast
import os

import numpy as np
import torch

from vit_amd.config import ViTConfig

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    cfg = ViTConfig(**ast.literal_eval(str(z["config_json"])))
    return cfg, int(z["batch"]), {k: z[k] for k in z.files}


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b| : the 'relative error' of BASELINE.json's fp32 bar."""
    if a.is_cuda and b.is_cuda and a.numel() > (1 << 24):
        # whole bench-shape matrices: the subtraction and the two maxima run where the data is (plain torch
        # elementwise ops in fp32 on exact bf16 / fp32 inputs; no 2 x 2.4 GB host copies per comparison)
        a, b = a.float(), b.float()
        return ((a - b).abs().max().double() / b.abs().max().double().clamp_min(1e-30)).item()
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def cosine(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return (a @ b / (a.norm() * b.norm())).item()


def min_row_cosine(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    a = a.reshape(-1, a.shape[-1])
    b = b.reshape(-1, b.shape[-1])
    return ((a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))).min().item()
