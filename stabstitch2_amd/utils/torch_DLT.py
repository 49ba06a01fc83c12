"""utils/torch_DLT.py of the reference: tensor_DLT(src_p, dst_p) -> H [B,3,3], solved in fp64 on device."""
from .. import ops


def tensor_DLT(src_p, dst_p):
    return ops.tensor_dlt(src_p, dst_p)
