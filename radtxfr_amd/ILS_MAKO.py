"""Drop-in for the reference's stand-alone ILS_MAKO.py (Gaussian MAKO line shape, ILS_MAKO.py:2-35).

Differs from radiative_transfer.ILS_MAKO exactly as the reference's two variants differ: Gaussian
weights with sigma = |gradient(X_out)| (no 1.6 factor), no clipping of bands to the input range,
no resFactor. The weights are evaluated on the GPU out to 14 sigma (beyond that exp(-z^2/2) is
below 3e-43 of the peak, invisible to the float32 sums)."""
import numpy as np

from .radiative_transfer import _MAKO_UM, _ils_apply, _is_torch


def ILS_MAKO(X, Y):
    """X (nX,) ascending wavenumbers, Y (nX,) or (nX,nS) -> (X_out (128,), Y_out (128,) or (128,nS))."""
    Xh = X.detach().cpu().numpy() if _is_torch(X) else np.asarray(X, dtype=np.float64)
    X_out = np.sort(10000.0 / _MAKO_UM)
    sigma_out = np.abs(np.gradient(X_out))
    return X_out, _ils_apply(1, Xh, Y, X_out, sigma_out)
