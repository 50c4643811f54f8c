"""MI355X-native Neural-CDE forward/adjoint integrator: a drop-in for the reference's
``torchcde.cdeint`` / ``src.ncde.NeuralCDE`` hot path.  Import as ``ncde_amd`` (see ncde_amd.py).

The compute path is hand-written HIP for gfx950 behind a C-ABI (include/ncde_hip.h); this package is
the thin Python host mirroring the reference's call surface.  There is no CPU fallback.
"""
from . import data  # noqa: F401
from ._lib import NcdeError, lib  # noqa: F401
from .interpolation import LinearInterpolation, NaturalCubicSpline  # noqa: F401
from .solver import FieldSpec, cdeint, coop_status  # noqa: F401
from .vector_fields import GRUGatedVectorField, MinimalGatedVectorField, MLPField, OriginalVectorField  # noqa: F401
from .ncde import NeuralCDE  # noqa: F401
from .coefficients import linear_interpolation_coeffs, natural_cubic_coeffs, natural_cubic_spline_coeffs  # noqa: F401
from .losses import MaskedTemporalLoss, RMSELoss, TemporalLossWrapper, masked_mean  # noqa: F401
