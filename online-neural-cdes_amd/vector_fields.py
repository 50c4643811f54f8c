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


class OriginalVectorField(nn.Module):
    def __init__(self, input_dim, hidden_dim, hidden_hidden_dim=15, num_layers=1, sparsity=None,
                 vector_field_type="matmul"):
        super().__init__()
        if vector_field_type != "matmul":
            raise NotImplementedError("only vector_field_type='matmul' is implemented")
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.hidden_hidden_dim, self.num_layers = hidden_hidden_dim, num_layers
        self.sparsity, self.vector_field_type = sparsity, vector_field_type
        self.output_dim = hidden_dim * input_dim
        self.nfe = 0
        first = nn.Linear(hidden_dim, hidden_hidden_dim)
        mods = [first, nn.ReLU()]
        if num_layers > 1:
            shared = nn.Linear(hidden_hidden_dim, hidden_hidden_dim)
            for _ in range(num_layers - 1):      # the SAME module object each time: weights are shared
                mods += [shared, nn.ReLU()]
        self.net_to_hh = nn.Sequential(*mods)
        self.tanh_output_layer = nn.Sequential(nn.Linear(hidden_hidden_dim, self.output_dim), nn.Tanh())

    def fused_spec(self):
        lins = [m for m in self.net_to_hh if isinstance(m, nn.Linear)]
        out = self.tanh_output_layer[0]
        return FieldSpec([(m.weight, m.bias) for m in lins], out.weight, out.bias)

    def forward(self, t, h):
        out = self.tanh_output_layer(self.net_to_hh(h)).view(-1, self.hidden_dim, self.input_dim)
        self.nfe += 1
        return out


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
