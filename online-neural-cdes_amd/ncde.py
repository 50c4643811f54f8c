"""``NeuralCDE``: drop-in for the reference model class (/root/reference/src/ncde/ncde.py:34-243).

Same constructor arguments, same ``state_dict`` layout (``initial_linear.*``, ``func.net_to_hh.*``,
``func.tanh_output_layer.0.*``, ``final_linear.*``) and same ``forward(inputs)`` contract; the solve
itself is one fused HIP kernel per direction (see solver.py).  The two tiny Linears outside the solve
stay ordinary torch modules.
"""
import torch
from torch import nn

from .interpolation import LinearInterpolation, NaturalCubicSpline
from .solver import cdeint
from .vector_fields import GRUGatedVectorField, MinimalGatedVectorField, OriginalVectorField

SPLINES = {
    "cubic": NaturalCubicSpline,
    "linear": LinearInterpolation,
    "rectilinear": LinearInterpolation,
}
VECTOR_FIELDS = {"original": OriginalVectorField, "minimal": MinimalGatedVectorField, "gru": GRUGatedVectorField}


class NeuralCDE(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, static_dim=None, hidden_hidden_dim=15, num_layers=3,
                 use_initial=True, interpolation="linear", interpolation_eps=None, sparsity=None,
                 vector_field="original", vector_field_type="matmul", adjoint=True, solver="rk4",
                 return_sequences=False, apply_final_linear=True, return_filtered_rectilinear=True,
                 kernel_flags=0):
        super().__init__()
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.static_dim = static_dim
        self.hidden_hidden_dim, self.num_layers = hidden_hidden_dim, num_layers
        self.use_initial = use_initial
        self.interpolation, self.interpolation_eps = interpolation, interpolation_eps
        self.sparsity = sparsity
        self.vector_field, self.vector_field_type = vector_field, vector_field_type
        self.adjoint, self.solver = adjoint, solver
        self.return_sequences = return_sequences
        self.apply_final_linear = apply_final_linear
        self.return_filtered_rectilinear = return_filtered_rectilinear
        self.kernel_flags = kernel_flags

        if self.initial_dim > 0:
            self.initial_linear = nn.Linear(self.initial_dim, hidden_dim)
        if interpolation in ("linear_cubic_smoothing", "linear_quintic_smoothing", "rectilinear_cubic_smoothing"):
            raise NotImplementedError("smoothed interpolation schemes are outside the fused path (SURVEY.md §2 row 9)")
        assert interpolation in SPLINES, "Unrecognised interpolation scheme {}".format(interpolation)
        assert interpolation_eps in (None, 1)
        self.spline = SPLINES[interpolation]
        # the reference asserts solver in ["rk4", "dopri5"] (ncde.py:129); the fixed-step family is what is fused
        assert solver in ("rk4", "dopri5", "midpoint", "euler")
        self.atol, self.rtol = 1e-5, 1e-3
        self.cdeint_options = {"min_step": 0.5} if solver == "dopri5" else {"step_size": 1}      # ncde.py:130-134
        if vector_field not in VECTOR_FIELDS:
            raise NotImplementedError("vector_field '%s' (sparse / low-rank) is outside the fused path" % vector_field)
        self.func = VECTOR_FIELDS[vector_field](input_dim=input_dim, hidden_dim=hidden_dim,
                                                hidden_hidden_dim=hidden_hidden_dim, num_layers=num_layers,
                                                sparsity=sparsity, vector_field_type=vector_field_type)
        self.final_linear = nn.Linear(hidden_dim, output_dim) if apply_final_linear else (lambda x: x)

    @property
    def initial_dim(self):
        d = self.input_dim if self.use_initial else 0
        if self.static_dim is not None:
            d += self.static_dim
        return d

    @property
    def nfe(self):
        return getattr(self.func, "nfe", None)

    def _control_and_start(self, inputs):
        """(control path X, z(t0)) from the module input: coefficients, or (static features, coefficients) when static_dim is set.
        The read-in layer sees [static features,] [X(t0)] -- whichever of the two the configuration provides (reference semantics:
        src/ncde/ncde.py:170-198) -- and with neither, z(t0) = 0."""
        if self.static_dim:
            if not (isinstance(inputs, (tuple, list)) and len(inputs) == 2):
                raise AssertionError("Inputs must be a 2-tuple of (static_data, temporal_data)")
            static, coeffs = inputs
        else:
            static, coeffs = None, inputs
        X = self.spline(coeffs)
        feats = ([static] if static is not None else []) + ([X.evaluate(0)] if self.use_initial else [])
        if not feats:
            return X, coeffs.new_zeros(coeffs.size(0), self.hidden_dim)
        return X, self.initial_linear(feats[0] if len(feats) == 1 else torch.cat(feats, dim=-1))

    def _readout(self, z):
        """z: the solution [B, n_times, H].  Final time only, or every time; on a rectilinear path every other knot is the lagged
        copy the rectilinear preparation inserted, dropped unless asked for (src/ncde/ncde.py:200-212)."""
        if not self.return_sequences:
            return self.final_linear(z[:, -1])
        y = self.final_linear(z)
        return y[:, ::2] if (self.interpolation == "rectilinear" and self.return_filtered_rectilinear) else y

    def forward(self, inputs):
        X, z0 = self._control_and_start(inputs)
        z = cdeint(X, self.func, z0, t=X.grid_points if self.return_sequences else X.interval, adjoint=self.adjoint,
                   vector_field_type=self.vector_field_type, method=self.solver, atol=self.atol, rtol=self.rtol,
                   options=dict(self.cdeint_options), kernel_flags=self.kernel_flags)
        return self._readout(z)
