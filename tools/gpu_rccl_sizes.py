#!/usr/bin/env python3
"""Lease script: does torch.distributed/RCCL move large buffers intact?  world 1, uint8 / int64 views, sizes around 2^31 and 2^32."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29722")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
for nbytes in (1 << 30, (1 << 31) - 8, (1 << 31) + 8, 2117800000, (1 << 32) + 64, 3 * (1 << 31)):
    src = torch.arange(nbytes // 8, dtype=torch.int64, device=dev)
    for dt in (torch.uint8, torch.int64):
        s = src.view(dt)
        for kind in ("a2a", "ag"):
            r = torch.zeros_like(s)
            if kind == "a2a":
                dist.all_to_all_single(r, s, output_split_sizes=[s.numel()], input_split_sizes=[s.numel()])
            else:
                dist.all_gather_into_tensor(r, s)
            torch.cuda.synchronize()
            ok = torch.equal(r, s)
            bad = 0 if ok else int((r.view(torch.int64) != src).sum().item())
            print("bytes %11d dtype %-12s %-3s %s bad_words %d" % (nbytes, str(dt), kind, "ok" if ok else "CORRUPT", bad), flush=True)
            del r
    del src
dist.destroy_process_group()
