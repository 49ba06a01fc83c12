"""utils/torch_tps_transform_point.py of the reference: transformer(point, source, target) -> [B,P,2]."""
from .. import ops


def transformer(point, source, target):
    return ops.tps_points(point, source, ops.tps_solve(source, target))
