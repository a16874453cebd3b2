"""Beam smoothing of sky maps: ``ProjectionMap.smooth`` (map/projection.py:485-504)."""

from __future__ import annotations

import numpy as np
import torch

from ._lib import Context, ptr


def smooth(data, weight=None, sigma=None, fwhm=None, x_res=1.0, y_res=1.0, device="cuda:0", ctx=None):
    """``numer = G(data*weight); denom = G(weight); data = where(denom > 0, numer/denom, 0)``
    over the last two axes, sigma in the map's angular units, resolutions per pixel.

    ``data`` (and ``weight``): [..., ny, nx] numpy arrays or device tensors.  Returns
    ``(smoothed, denom)`` of the input kind; ``denom`` is the reference's new weight.
    """
    if not (sigma is None) ^ (fwhm is None):
        raise ValueError("You must supply exactly one of 'sigma' or 'fwhm'.")
    sigma = sigma if sigma is not None else fwhm / np.sqrt(8 * np.log(2))
    x_sigma_pixels, y_sigma_pixels = abs(sigma / x_res), abs(sigma / y_res)
    as_numpy = not isinstance(data, torch.Tensor)
    dev = torch.device(device)
    d = torch.as_tensor(np.ascontiguousarray(data, np.float32)).to(dev) if as_numpy else data.to(dev, torch.float32).contiguous()
    w = None
    if weight is not None:
        w = torch.as_tensor(np.array(np.broadcast_to(weight, data.shape), dtype=np.float32)).to(dev) if as_numpy else weight.to(dev, torch.float32).expand_as(d).contiguous()
    torch.cuda.set_device(dev)
    ctx = ctx or Context(dev.index or 0)
    ctx.set_stream(torch.cuda.current_stream(dev))
    ny, nx = d.shape[-2:]
    out, den = torch.empty_like(d), torch.empty_like(d)
    tmp = torch.empty(2 * ny * nx, dtype=torch.float32, device=dev)
    flat_d, flat_o, flat_n = d.reshape(-1, ny, nx), out.reshape(-1, ny, nx), den.reshape(-1, ny, nx)
    flat_w = None if w is None else w.reshape(-1, ny, nx)
    for k in range(flat_d.shape[0]):
        ctx.call(
            "mrx_map_smooth", ptr(flat_d[k]), ptr(None if flat_w is None else flat_w[k]), ptr(flat_o[k]),
            ptr(flat_n[k]), ptr(tmp), ny, nx, float(y_sigma_pixels), float(x_sigma_pixels),
        )
    if as_numpy:
        return out.cpu().numpy(), den.cpu().numpy()
    return out, den
