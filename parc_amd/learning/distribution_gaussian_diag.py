"""Diagonal Gaussian action head (mirror of ``distribution_gaussian_diag.py:8-103``; parameter names ``_mean_net``,
``_logstd_net`` kept so checkpoints interchange)."""
import enum

import numpy as np
import torch


class StdType(enum.Enum):
    FIXED = 0
    CONSTANT = 1
    VARIABLE = 2


class DistributionGaussianDiagBuilder(torch.nn.Module):
    def __init__(self, in_size, out_size, std_type, init_std, init_output_scale=0.01):
        super().__init__()
        self._std_type = std_type
        self._mean_net = torch.nn.Linear(in_size, out_size)
        torch.nn.init.uniform_(self._mean_net.weight, -init_output_scale, init_output_scale)
        torch.nn.init.zeros_(self._mean_net.bias)
        logstd = float(np.log(init_std))
        if std_type in (StdType.FIXED, StdType.CONSTANT):
            grad = std_type == StdType.CONSTANT
            self._logstd_net = torch.nn.Parameter(torch.full([out_size], logstd, dtype=torch.float32), requires_grad=grad)
        elif std_type == StdType.VARIABLE:
            self._logstd_net = torch.nn.Linear(in_size, out_size)
            torch.nn.init.uniform_(self._logstd_net.weight, -init_output_scale, init_output_scale)
            torch.nn.init.constant_(self._logstd_net.bias, logstd)
        else:
            raise AssertionError("Unsupported StdType: {}".format(std_type))

    def forward(self, x):
        mean = self._mean_net(x)
        if self._std_type == StdType.VARIABLE:
            logstd = self._logstd_net(x)
        else:
            logstd = torch.broadcast_to(self._logstd_net, mean.shape)
        return DistributionGaussianDiag(mean=mean, logstd=logstd)


class DistributionGaussianDiag:
    def __init__(self, mean, logstd):
        self._mean = mean
        self._logstd = logstd
        self._std = torch.exp(logstd)
        self._dim = mean.shape[-1]

    stddev = property(lambda self: self._std)
    logstd = property(lambda self: self._logstd)
    mean = property(lambda self: self._mean)
    mode = property(lambda self: self._mean)

    def sample(self):
        return self._mean + self._std * torch.randn_like(self._mean)

    def log_prob(self, x):
        diff = x - self._mean
        logp = -0.5 * torch.sum(torch.square(diff / self._std), dim=-1)
        logp += -0.5 * self._dim * np.log(2.0 * np.pi) - torch.sum(self._logstd, dim=-1)
        return logp

    def entropy(self):
        return torch.sum(self._logstd, dim=-1) + 0.5 * self._dim * np.log(2.0 * np.pi * np.e)

    def kl(self, other):
        other_var = torch.square(other.stddev)
        res = torch.sum(other.logstd - self._logstd + (torch.square(self._std) + torch.square(self._mean - other.mean)) / (2.0 * other_var), dim=-1)
        return res - 0.5 * self._dim

    def param_reg(self):
        return torch.sum(torch.square(self._mean), dim=-1)
