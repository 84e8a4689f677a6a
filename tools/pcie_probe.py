"""Raw device-to-host rate of this box (pinned and pageable), for reading
tools/trace_probe.py against.  torch is used for the buffers only."""
import json
import time

import torch

dev = torch.device("cuda", 0)
out = {}
for mb in (4, 32, 512):
    n = mb << 20
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    for kind in ("pinned", "pageable"):
        dst = torch.empty(n, dtype=torch.uint8, pin_memory=(kind == "pinned"))
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        reps = max(2, 2048 // mb // 4)
        t0 = time.perf_counter()
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out[f"{kind}_{mb}MiB_GBps"] = n * reps / el / 1e9
print(json.dumps(out))
