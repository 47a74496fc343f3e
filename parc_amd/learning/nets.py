"""MLP trunks by name (``PARC/util/nets/net_builder.py:6``; default ``fc_3layers_2048units`` = 2048-1024-512 ReLU,
zero biases, ``fc_3layers_2048units.py:5``)."""
import numpy as np
import torch

_LAYERS = {
    "fc_3layers_2048units": [2048, 1024, 512],
    "fc_3layers_1024units": [1024, 1024, 512],
    "fc_2layers_1024units": [1024, 512],
    "fc_2layers_512units": [512, 256],
    "fc_2layers_128units": [128, 64],
    "fc_1layers_16units": [16],
}


def build_net(net_name, input_dict, activation=torch.nn.ReLU):
    if net_name not in _LAYERS:
        raise AssertionError("Unsupported net: {}".format(net_name))
    in_size = int(np.sum([np.prod(v.shape) for v in input_dict.values()]))
    layers = []
    for out_size in _LAYERS[net_name]:
        lin = torch.nn.Linear(in_size, out_size)
        torch.nn.init.zeros_(lin.bias)
        layers += [lin, activation()]
        in_size = out_size
    return torch.nn.Sequential(*layers), dict()


def calc_layers_out_size(layers):
    for m in reversed(list(layers.modules())):
        if isinstance(m, torch.nn.Linear):
            return m.out_features
    raise AssertionError("no linear layer")
