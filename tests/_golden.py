"""Helpers shared by the parity tests: load the committed reference fixtures."""
import os

import numpy as np
from scipy import sparse

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ELASTICITY_CASES = {
    # name: element type
    "ps32_quad": "quad",
    "ps32_quad_ramp": "quad",
    "ps13x9_quad": "quad",
    "ps13x9_quad_distorted": "quad",
    "ps11x8_tri": "tri",
    "ps11x8_tri_ramp": "tri",
    "arm_tri": "tri",
    "ps3d_6x4x3_block": "block",
    "ps3d_5x4x4_block_ramp_distorted": "block",
}
HELMHOLTZ_CASES = {"helm32_quad": "quad", "helm12x9_tri": "tri"}


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def csr_of(g, prefix, n):
    return sparse.csr_matrix((g[f"{prefix}_data"], g[f"{prefix}_indices"], g[f"{prefix}_indptr"]), shape=(n, n))


def force_dict(g):
    return {int(n): [float(x) for x in v] for n, v in zip(g["force_nodes"], g["force_vals"])}


def relmax(a, b):
    """max|a-b| / max|b|  -- the reference's own K-parity convention
    (examples/SciTech2023/verification/verify.py:37-38)."""
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())
