"""
Plumbing shared by the operator-level mirrors: host array <-> device tensor conversion around one
C-ABI call.  NumPy in -> NumPy out (copies over PCIe); torch CUDA tensor in -> tensor out (stays resident).
"""
import numpy as np

from ._native import Context, ptr, to_host


def is_tensor(x):
    try:
        import torch
    except ImportError:  # pragma: no cover
        return False
    return isinstance(x, torch.Tensor)


class Staged(object):
    """Inputs moved to the device for one call; remembers whether results go back to the host."""

    def __init__(self, *arrays, **kw):
        self.ctx = Context.current(kw.get('device'))
        self.on_device = any(is_tensor(a) for a in arrays if a is not None)

    def inp(self, array, dtype=np.float64):
        return self.ctx.to_device(array, dtype)

    def out(self, shape, dtype=None):
        return self.ctx.empty(shape, dtype)

    def result(self, tensor, shape=None, dtype=None):
        if self.on_device:
            return tensor if shape is None else tensor.reshape(shape)
        return to_host(tensor, dtype=dtype, shape=shape)


__all__ = ['Staged', 'ptr', 'is_tensor']
