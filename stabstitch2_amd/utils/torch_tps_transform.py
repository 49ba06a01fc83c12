"""utils/torch_tps_transform.py of the reference: transformer(U, source, target, out_size, mode).
source = warped mesh (canvas-normalised), target = rigid mesh (image-normalised): backward map."""
from .. import ops


def transformer(U, source, target, out_size, mode='NORMAL'):
    T = ops.tps_solve(source, target)
    return ops.tps_warp(U, source, T, int(out_size[0]), int(out_size[1]), mode)
