"""SGD / AdamW wrapper (``optimizer.py:4-46``) + the one thing the reference does not have: a bucketed gradient
all-reduce over RCCL before the parameter update when more than one rank trains."""
import torch

from parc_amd.learning import dist_util


class Optimizer:
    def __init__(self, config, param_list):
        self._param_list = param_list
        lr = float(config["learning_rate"])
        wd = float(config.get("weight_decay", 0.0))
        t = config["type"]
        if t == "SGD":
            self._optimizer = torch.optim.SGD(param_list, lr, momentum=0.9, weight_decay=wd)
        elif t == "Adam":
            self._optimizer = torch.optim.AdamW(param_list, lr, weight_decay=wd)
        else:
            raise AssertionError("Unsupported optimizer type: " + t)
        self._steps = 0
        self._bucket = dist_util.GradBucket(param_list) if dist_util.is_dist() else None
        self.sync()

    def step(self, loss, **kwargs):
        self._optimizer.zero_grad()
        loss.backward()
        if self._bucket is not None:
            self._bucket.all_reduce_grads()
        if "model" in kwargs:
            max_norm = kwargs["max_norm"]
            grad_norm = torch.nn.utils.clip_grad_norm_(kwargs["model"].parameters(), max_norm, 2)
            if grad_norm.item() > max_norm:
                print("clipped grad norm:", grad_norm.item())
        self._optimizer.step()
        self._steps += 1

    def get_steps(self):
        return self._steps

    def sync(self):
        """Make every rank start from rank 0's parameters."""
        if dist_util.is_dist():
            for p in self._param_list:
                dist_util.broadcast_(p.data, 0)
