"""Rollout storage ``[steps, envs, ...]`` with shuffled minibatch sampling (mirror of ``experience_buffer.py:5-115``)."""
import torch


class ExperienceBuffer:
    def __init__(self, buffer_length, batch_size, device):
        self._buffer_length = buffer_length
        self._batch_size = batch_size
        self._device = device
        self._buffer_head = 0
        self._total_samples = 0
        self._buffers = dict()
        self._flat_buffers = dict()
        self._sample_buf = torch.randperm(buffer_length * batch_size, device=device, dtype=torch.long)
        self._sample_buf_head = 0

    def add_buffer(self, name, buffer):
        assert len(buffer.shape) >= 2 and buffer.shape[0] == self._buffer_length and buffer.shape[1] == self._batch_size
        assert name not in self._buffers
        self._buffers[name] = buffer
        self._flat_buffers[name] = buffer.view([buffer.shape[0] * buffer.shape[1]] + list(buffer.shape[2:]))

    def reset(self):
        self._buffer_head = 0
        self._reset_sample_buf()

    def clear(self):
        self.reset()
        self._total_samples = 0

    def inc(self):
        self._buffer_head = (self._buffer_head + 1) % self._buffer_length
        self._total_samples += self._batch_size

    def get_total_samples(self):
        return self._total_samples

    def get_sample_count(self):
        return min(self._total_samples, self._buffer_length * self._batch_size)

    def record(self, name, data):
        assert data.shape[0] == self._batch_size
        self._buffers[name][self._buffer_head] = data

    def get_buffer_head(self):
        return self._buffer_head

    def get_data(self, name):
        return self._buffers[name]

    def get_data_flat(self, name):
        return self._flat_buffers[name]

    def set_data(self, name, data):
        buf = self._buffers[name]
        assert buf.shape[0] == data.shape[0] and buf.shape[1] == data.shape[1]
        buf[:] = data

    def sample(self, n):
        """experience_buffer.py:81-89.  On the GPU the rows of ALL buffers are gathered by one HIP launch (parc_gather_rows: byte-exact,
        i.e. bit-identical to the per-buffer indexing kernels it replaces); CPU tensors (the gloo tests) keep torch's indexing."""
        idx = self._sample_rand_idx(n, wrap=False)
        count = self.get_sample_count()
        bufs = self._flat_buffers
        if len(bufs) == 0 or not all(v.is_cuda and v.is_contiguous() for v in bufs.values()) or len(bufs) > 16:
            idx = torch.remainder(idx, count)
            return {k: v[idx] for k, v in bufs.items()}
        import ctypes as C
        from parc_amd import lib as L
        lib = L.load()
        names = list(bufs)
        out = {k: torch.empty((n,) + tuple(bufs[k].shape[1:]), dtype=bufs[k].dtype, device=bufs[k].device) for k in names}
        nb = len(names)
        src = (C.c_void_p * nb)(*[bufs[k].data_ptr() for k in names])
        dst = (C.c_void_p * nb)(*[out[k].data_ptr() for k in names])
        rb = (C.c_int64 * nb)(*[(bufs[k][0].numel() * bufs[k].element_size()) for k in names])
        idx = idx.contiguous()
        L.check(lib.parc_gather_rows(nb, src, dst, rb, idx.data_ptr(), int(n), int(count), torch.cuda.current_stream().cuda_stream))
        return out

    def _reset_sample_buf(self):
        self._sample_buf[:] = torch.randperm(self._buffer_length * self._batch_size, device=self._device, dtype=torch.long)
        self._sample_buf_head = 0

    def _sample_rand_idx(self, n, wrap=True):
        total = self._sample_buf.shape[0]
        assert n <= total
        if self._sample_buf_head + n <= total:
            idx = self._sample_buf[self._sample_buf_head:self._sample_buf_head + n]
            self._sample_buf_head += n
        else:
            head = self._sample_buf[self._sample_buf_head:].clone()
            rem = n - head.shape[0]
            self._reset_sample_buf()
            idx = torch.cat([head, self._sample_buf[:rem]], dim=0)
            self._sample_buf_head = rem
        return torch.remainder(idx, self.get_sample_count()) if wrap else idx   # (the gather kernel takes the remainder itself)
