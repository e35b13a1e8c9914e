"""Array conventions shared by the host-side mirror.

The reference passes Julia `Matrix{Float64}` of shape (D, T), column-major.  The mirror takes numpy arrays
of the same SHAPE (D, T); they are converted to Fortran order (no copy when already so), which is byte for
byte the Julia memory image the C-ABI expects.  Device-resident inputs are torch tensors on the current HIP
device with the same logical shape (D, T) and unit stride along D (e.g. `buf.view(T, ld)[:, :D].t()`).
"""
import numpy as np


def jl_matrix(a, name="array"):
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError(f"{name} must be a matrix, got ndim={a.ndim}")
    return np.asfortranarray(a, dtype=np.float64)


def jl_vector(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


def is_torch(x):
    return type(x).__module__.split(".")[0] == "torch"


def dev_matrix(x, name="tensor"):
    """Return (data_ptr, D, T, ld) of a (D,T) torch tensor with unit stride along D on a HIP device."""
    import torch

    if x.dtype != torch.float64:
        raise TypeError(f"{name} must be float64")
    if not x.is_cuda:
        raise ValueError(f"{name} must live on the HIP device")
    if x.dim() != 2:
        raise ValueError(f"{name} must be a matrix")
    D, T = x.shape
    if D > 1 and x.stride(0) != 1:
        raise ValueError(f"{name} must have unit stride along its first (feature) axis: pass buf.view(T, ld).t()")
    ld = x.stride(1) if T > 1 else max(D, 1)
    return x.data_ptr(), D, T, ld


def current_stream_ptr():
    import torch

    return torch.cuda.current_stream().cuda_stream
