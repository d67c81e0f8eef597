"""`SAVGOLFilter` with the reference's constructor and call signature (reference: savgol_filer.py:7-70), the
smoothing itself on the mvx HIP kernel.

The reference runs scipy.signal.savgol_filter on the host, coordinate by coordinate; here a (T, K, C) pose
tensor is filtered in one launch.  numpy arrays and CPU tensors are uploaded, filtered on the current HIP
device and returned in the type they came in, as the reference returns them; there is no host filter.

Precision: the kernel reads float32 samples, sums each window in float64 and rounds the result once to float32.
float64 input (which scipy filters in float64) is therefore rounded to float32 on the way in and comes back as
float64 values that carry float32 precision: about 6e-8 relative, 2.5e-4 mm on coordinates of a few metres.
"""
import time
import warnings

import numpy as np
import torch

from poseestimation_amd import _cabi

_HAT_CACHE = {}


def fit_window(window_size, n_steps, polyorder=2):
    """The window and order the reference ends up with (savgol_filer.py:33-45): an even window loses one sample,
    a window longer than the sequence is cut to it, and the order drops below a very short window."""
    window = window_size - 1 if window_size % 2 == 0 else window_size
    if window > n_steps:
        window = n_steps
    order = window - 1 if window <= polyorder else polyorder
    assert order > 0
    assert window > order
    return window, order


def hat_matrix(window, order, device):
    """window x window float64 matrix A pinv(A) of the least-squares polynomial fit on a window, positions
    centred for conditioning (the fitted values do not depend on the origin); cached per device."""
    key = (window, order, str(device))
    if key not in _HAT_CACHE:
        pos = np.arange(window, dtype=np.float64) - (window - 1) / 2.0
        A = np.vander(pos, order + 1)
        H = A @ np.linalg.pinv(A)
        _HAT_CACHE[key] = torch.from_numpy(np.ascontiguousarray(H)).to(device)
    return _HAT_CACHE[key]


def savgol_device(x, window, order):
    """x (T, ...) float32 on a HIP device -> same shape, filtered along dim 0."""
    x = _cabi.device_f32(x, "x")
    n_steps = x.shape[0]
    n_series = x.numel() // n_steps if n_steps else 0
    out = torch.empty_like(x)
    hat = hat_matrix(window, order, x.device)
    with torch.cuda.device(x.device):
        st = _cabi.load().mvx_savgol_filter(_cabi.ptr(x), n_steps, n_series, _cabi.ptr(hat), window, _cabi.ptr(out),
                                            _cabi.stream(x.device))
    _cabi.check(st, "mvx_savgol_filter")
    return out


class SAVGOLFilter:
    """1-D Savitzky-Golay filter along time, polynomial order 2.  `filter(x)` with x (T, K, C) returns
    (smoothed poses in x's type and place, seconds spent)."""

    def __init__(self, window_size):
        self.window_size = window_size
        self.polyorder = 2

    def __call__(self, x=None):
        window, order = fit_window(self.window_size, x.shape[0], self.polyorder)
        if len(x.shape) != 3:
            warnings.warn('x should be a tensor or numpy of [T*M,K,C]')
        assert len(x.shape) == 3
        start = time.time()
        if isinstance(x, torch.Tensor):
            if x.dtype not in (torch.float32, torch.float64):
                x = x.double()                # scipy filters other dtypes as float64
            dev = x if x.is_cuda else x.cuda()
            smooth = savgol_device(dev, window, order).to(dtype=x.dtype)
            if not x.is_cuda:
                smooth = smooth.cpu()
        else:
            arr = np.asarray(x)
            kind = arr.dtype if arr.dtype in (np.float32, np.float64) else np.float64
            smooth = savgol_device(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).cuda(), window, order)
            smooth = smooth.cpu().numpy().astype(kind)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        return smooth, time.time() - start
