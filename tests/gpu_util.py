"""Helpers for the -m gpu tests: everything goes through the C ABI of libsot_hip.so (via the
package's ctypes binding); a missing library is a hard failure, never a skip or a fallback."""
import os

import numpy as np
import torch


def device():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def native():
    import sot_amd
    lib_path = sot_amd._native.library_path()
    assert os.path.exists(lib_path), f"{lib_path} missing on a GPU box: the HIP path must be built in-tree"
    sot_amd._native.load(build_if_missing=False)
    return sot_amd._native


def module_for(ctor):
    from sot_amd.losses import Wasserstein1D
    native()
    return Wasserstein1D(**ctor).to(device())


def to_dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(device())


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)
