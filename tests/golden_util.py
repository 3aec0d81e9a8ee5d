"""Helpers shared by the parity tests: load a golden fixture and rebuild its parameters from the seed."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
if GOLD not in sys.path:
    sys.path.insert(0, GOLD)
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import params as GP  # noqa: E402


def load_case(tag):
    z = np.load(os.path.join(GOLD, tag + ".npz"))
    cfg = json.loads(str(z["cfg_json"]))
    shapes = [(k, tuple(s)) for k, s in json.loads(str(z["shapes_json"]))]
    names = json.loads(str(z["grad_names_json"]))
    return z, cfg, shapes, names


def build_state(shapes, seed, kind, T=None, res=None, prefix=None, state_fn=None):
    """{key: fp32 tensor} regenerated from the seed (+ the integer index buffers the oracle reads)."""
    import oracle.swin as OS
    P = (state_fn or GP.seeded_state)(shapes, seed)
    if kind == "swin_block":
        P = {"blk." + k: v for k, v in P.items()}
        ws, _ = OS.block_geometry(res, res, 7, 0)
        P["blk.attn.relative_position_index"] = OS.relative_position_index(ws)
        P["blk.attn.t_relative_coords"] = OS.temporal_relative_index(T)
        P["blk.attn.t_relative_coords_a"] = OS.temporal_relative_index(T)
    elif kind == "swin":
        for k in list(P):
            if k.endswith("attn.relative_position_bias_table"):
                pre = k[: -len("relative_position_bias_table")]
                P[pre + "relative_position_index"] = OS.relative_position_index(7)
                P[pre + "t_relative_coords"] = OS.temporal_relative_index(T)
                P[pre + "t_relative_coords_a"] = OS.temporal_relative_index(T)
    elif kind == "vit_block":
        P = {"blk." + k: v for k, v in P.items()}
    return P
