#!/usr/bin/env python3
"""Developer tool (GPU box): the practical HBM peak of the box -- device-to-device copy and fill bandwidth with torch (SURVEY 8(d):
"report achieved-copy BW as the practical peak too").  Prints one JSON line."""
import json, torch
dev = "cuda:0"
n = 1 << 29  # 2 GiB of fp32 per buffer: beyond the 256 MB Infinity Cache
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
t_copy = timed(lambda: b.copy_(a))
t_fill = timed(lambda: b.fill_(1.0))
t_read = timed(lambda: a.sum())
gb = n * 4 / 1e9
print(json.dumps({"copy_GBps_read_plus_write": round(2 * gb / t_copy, 1), "fill_GBps_write": round(gb / t_fill, 1), "sum_GBps_read": round(gb / t_read, 1),
                  "buffer_GB": round(gb, 2), "device": torch.cuda.get_device_name(0)}))
