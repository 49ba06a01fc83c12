"""utils/torch_homo_transform.py of the reference: transformer(U, theta, out_size) on NCHW tensors."""
from .. import ops


def transformer(U, theta, out_size, **kwargs):
    b = U.shape[0]
    return ops.homo_warp_nchw(U.float().contiguous(), theta.reshape(b, 3, 3).float().contiguous(),
                              int(out_size[0]), int(out_size[1]))
