"""Vector fields f_theta: R^H -> R^{H x C} the fused kernels understand.

``OriginalVectorField`` mirrors the reference's class of the same name
(/root/reference/src/ncde/vector_fields/base.py:7-104): same constructor, same ``state_dict`` keys
(``net_to_hh.{0,2,4,...}`` -- every index >= 2 is ONE shared Linear -- and ``tanh_output_layer.0``), same
``nfe`` counter, and a plain-torch ``forward(t, h)`` for use outside ``cdeint``.
``MLPField`` is the un-shared general stack (the toy's CDEFunc, experiments/sim_bm_toy_example.py:10-30).
"""
import torch
from torch import nn

from .solver import FieldSpec


class BaseVectorField(nn.Module):
    """Common part of the reference's fields (base.py:7-92): the inner net H (or H+C) -> HH -> ... -> HH with ONE
    Linear shared by all inner layers, ``nfe``, and the [H, C] view of the output in the matmul mode."""

    def __init__(self, input_dim, hidden_dim, hidden_hidden_dim=15, num_layers=1, sparsity=None,
                 vector_field_type="matmul"):
        super().__init__()
        if vector_field_type not in ("matmul", "evaluate", "derivative"):
            raise ValueError("vector_field_type string not recognised")
        if sparsity is not None:
            raise NotImplementedError("sparse / low-rank fields (third-party sparselinear) are outside the fused path")
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.hidden_hidden_dim, self.num_layers = hidden_hidden_dim, num_layers
        self.sparsity, self.vector_field_type = sparsity, vector_field_type
        self.matmul = vector_field_type == "matmul"
        self.initial_dim = hidden_dim if self.matmul else hidden_dim + input_dim
        self.output_dim = hidden_dim * input_dim if self.matmul else hidden_dim
        self.nfe = 0
        first = nn.Linear(self.initial_dim, hidden_hidden_dim)
        mods = [first, nn.ReLU()]
        if num_layers > 1:
            shared = nn.Linear(hidden_hidden_dim, hidden_hidden_dim)
            for _ in range(num_layers - 1):      # the SAME module object each time: weights are shared
                mods += [shared, nn.ReLU()]
        self.net_to_hh = nn.Sequential(*mods)
        self.additional_network_initialisation()

    def _inner_layers(self):
        return [(m.weight, m.bias) for m in self.net_to_hh if isinstance(m, nn.Linear)]

    def forward(self, t, h):
        out = self._forward(h)
        if self.matmul:
            out = out.view(-1, self.hidden_dim, self.input_dim)
        self.nfe += 1
        return out


class OriginalVectorField(BaseVectorField):
    def additional_network_initialisation(self):
        self.tanh_output_layer = nn.Sequential(nn.Linear(self.hidden_hidden_dim, self.output_dim), nn.Tanh())

    def fused_spec(self):
        out = self.tanh_output_layer[0]
        return FieldSpec(self._inner_layers(), out.weight, out.bias, "original", self.vector_field_type)

    def _forward(self, h):
        return self.tanh_output_layer(self.net_to_hh(h))


class MinimalGatedVectorField(BaseVectorField):
    """sigmoid(Linear_z(hh)) * tanh(Linear_r(hh)) (gating.py:7-32); same ``state_dict`` keys as the reference."""

    def additional_network_initialisation(self):
        self.sigmoid_net = nn.Sequential(nn.Linear(self.hidden_hidden_dim, self.output_dim), nn.Sigmoid())
        self.tanh_net = nn.Sequential(nn.Linear(self.hidden_hidden_dim, self.output_dim), nn.Tanh())

    def fused_spec(self):
        sg, th = self.sigmoid_net[0], self.tanh_net[0]
        return FieldSpec(self._inner_layers(), th.weight, th.bias, "minimal", self.vector_field_type, sg.weight, sg.bias)

    def _forward(self, h):
        hh = self.net_to_hh(h)
        return self.sigmoid_net(hh) * self.tanh_net(hh)


class GRUGatedVectorField(BaseVectorField):
    """sigmoid_net(net(h)) * tanh_net(net(reset_net(h) * h)) (gating.py:35-61)."""

    def additional_network_initialisation(self):
        self.reset_net = nn.Sequential(nn.Linear(self.initial_dim, self.initial_dim), nn.Sigmoid())
        self.sigmoid_net = nn.Sequential(nn.Linear(self.hidden_hidden_dim, self.output_dim), nn.Sigmoid())
        self.tanh_net = nn.Sequential(nn.Linear(self.hidden_hidden_dim, self.output_dim), nn.Tanh())

    def fused_spec(self):
        rs, sg, th = self.reset_net[0], self.sigmoid_net[0], self.tanh_net[0]
        return FieldSpec(self._inner_layers(), th.weight, th.bias, "gru", self.vector_field_type, sg.weight, sg.bias,
                         rs.weight, rs.bias)

    def _forward(self, h):
        inner = self.net_to_hh(h)
        reset = self.net_to_hh(self.reset_net(h) * h)
        return self.sigmoid_net(inner) * self.tanh_net(reset)


class MLPField(nn.Module):
    """Linear+ReLU stack of arbitrary widths, then Linear+tanh viewed as [H, C]."""

    def __init__(self, input_dim, hidden_dim, widths):
        super().__init__()
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        dims = [hidden_dim] + list(widths)
        self.hidden_layers = nn.ModuleList(nn.Linear(dims[i], dims[i + 1]) for i in range(len(widths)))
        self.out_layer = nn.Linear(dims[-1], hidden_dim * input_dim)
        self.nfe = 0

    def fused_spec(self):
        return FieldSpec([(m.weight, m.bias) for m in self.hidden_layers], self.out_layer.weight, self.out_layer.bias)

    def forward(self, t, h):
        for m in self.hidden_layers:
            h = torch.relu(m(h))
        self.nfe += 1
        return torch.tanh(self.out_layer(h)).view(-1, self.hidden_dim, self.input_dim)
