"""Running mean/std normalizer (mirror of ``PARC/motion_tracker/learning/normalizer.py:8-119``): same state-dict
entries (``_count``, ``_mean``, ``_std`` as non-grad Parameters, so reference checkpoints load), same
record/update arithmetic.  New: the pending (count, sum, sum_sq) are all-reduced across ranks in ``update``."""
import numpy as np
import torch

from parc_amd.learning import dist_util


class Normalizer(torch.nn.Module):
    def __init__(self, shape, device, init_mean=None, init_std=None, min_std=1e-4, clip=np.inf, dtype=torch.float,
                 non_norm_indices=None):
        super().__init__()
        self._min_var = min_std * min_std
        self._clip = clip
        self.dtype = dtype
        self._non_norm_indices = non_norm_indices
        P = torch.nn.Parameter
        self._count = P(torch.zeros([1], device=device, dtype=torch.long), requires_grad=False)
        self._mean = P(torch.zeros(shape, device=device, dtype=dtype), requires_grad=False)
        self._std = P(torch.ones(shape, device=device, dtype=dtype), requires_grad=False)
        if init_mean is not None:
            assert init_mean.shape == tuple(self._mean.shape)
            self._mean[:] = init_mean
        if init_std is not None:
            assert init_std.shape == tuple(self._std.shape)
            self._std[:] = init_std
        self._mean_sq = None
        self._new_count = 0
        self._new_sum = torch.zeros_like(self._mean)
        self._new_sum_sq = torch.zeros_like(self._mean)

    def record(self, x):
        shape = self.get_shape()
        assert len(x.shape) > len(shape)
        x = x.flatten(start_dim=0, end_dim=len(x.shape) - len(shape) - 1)
        self._new_count += x.shape[0]
        self._new_sum += torch.sum(x, axis=0)
        self._new_sum_sq += torch.sum(torch.square(x), axis=0)

    def update(self):
        if self._mean_sq is None:
            self._mean_sq = self._calc_mean_sq(self._mean, self._std)
        if dist_util.is_dist():  # every rank contributes its shard's sums
            cnt = torch.tensor([float(self._new_count)], device=self._mean.device, dtype=torch.float64)
            dist_util.all_reduce_sum_(cnt)
            dist_util.all_reduce_sum_(self._new_sum)
            dist_util.all_reduce_sum_(self._new_sum_sq)
            self._new_count = int(cnt.item())
        new_count = self._new_count
        if new_count == 0:
            return
        new_mean = self._new_sum / new_count
        new_mean_sq = self._new_sum_sq / new_count
        new_total = self._count + new_count
        w_old = self._count.type(torch.float) / new_total.type(torch.float)
        w_new = float(new_count) / new_total.type(torch.float)
        self._mean[:] = w_old * self._mean + w_new * new_mean
        self._mean_sq[:] = w_old * self._mean_sq + w_new * new_mean_sq
        self._count[:] = new_total
        self._std[:] = self._calc_std(self._mean, self._mean_sq)
        self._new_count = 0
        self._new_sum[:] = 0
        self._new_sum_sq[:] = 0
        if self._non_norm_indices is not None:
            self._mean[self._non_norm_indices] = 0.0
            self._std[self._non_norm_indices] = 1.0

    def get_shape(self):
        return self._mean.shape

    def get_count(self):
        return self._count

    def get_mean(self):
        return self._mean

    def get_std(self):
        return self._std

    def set_mean_std(self, mean, std):
        assert mean.shape == self.get_shape() and std.shape == self.get_shape()
        self._mean[:] = mean
        self._std[:] = std
        self._mean_sq = self._calc_mean_sq(self._mean, self._std)

    def normalize(self, x):
        norm_x = (x - self._mean) / self._std
        norm_x = torch.clamp(norm_x, -self._clip, self._clip)
        return norm_x.type(self.dtype)

    def normalize_and_record(self, x, copy_out=None):
        """``normalize(x)`` and, in the same pass, ``copy_out[...] = x`` (the rollout buffer slot).  On a GPU this is one
        kernel of the env library (``parc_normalize_record``, bit-identical to ``normalize``); otherwise the torch ops."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2 and
                (copy_out is None or (copy_out.is_contiguous() and copy_out.shape == x.shape and copy_out.dtype == torch.float32))):
            if copy_out is not None:
                copy_out.copy_(x)
            return self.normalize(x)
        from parc_amd import lib as L
        lib = L.load()
        out = torch.empty_like(x)
        L.check(lib.parc_normalize_record(x.data_ptr(), self._mean.data_ptr(), self._std.data_ptr(), float(self._clip), out.data_ptr(),
                                          None if copy_out is None else copy_out.data_ptr(), x.shape[0], x.shape[1],
                                          torch.cuda.current_stream().cuda_stream))
        return out

    def unnormalize(self, norm_x):
        return (norm_x * self._std + self._mean).type(self.dtype)

    def _calc_std(self, mean, mean_sq):
        var = torch.clamp_min(mean_sq - torch.square(mean), self._min_var)
        return torch.sqrt(var).type(self.dtype)

    def _calc_mean_sq(self, mean, std):
        return (torch.square(std) + torch.square(mean)).type(self.dtype)
