"""Control-mesh resolution shared by every stage (the reference keeps the same two names in
Full_model_inference/Codes/grid_res.py:2-3; the HIP kernels hard-code the derived sizes in csrc/common.h)."""

GRID_H = 6                                   # mesh cells vertically   -> 7 vertex rows
GRID_W = 8                                   # mesh cells horizontally -> 9 vertex columns

NUM_VERTICES = (GRID_H + 1) * (GRID_W + 1)   # 63 TPS control points per view (SS_NV)
TPS_COEFFS = NUM_VERTICES + 3                # 66 spline coefficients per coordinate (SS_NT)
MESH_FLOATS = NUM_VERTICES * 2               # 126 = size of the regressors' last FC layer


def check_against_library():
    """The kernels are compiled for a 7 x 9 mesh; fail loudly if someone edits the numbers above."""
    if (GRID_H, GRID_W) != (6, 8):
        raise RuntimeError('libstabstitch_hip.so is built for GRID_H=6, GRID_W=8 (csrc/common.h: SS_GRID_H/W)')
