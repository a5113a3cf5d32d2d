#!/usr/bin/env python3
"""Does an all_to_all_single message beyond 4 GB arrive intact?  (round-2 note in kreeq_amd/dist.py: "messages beyond 4 GB
lost records" -- cause unknown.)  World size 1 under torchrun, backend nccl (= RCCL): every message goes to the rank itself
through the collective, with explicit split sizes like ShardedCounter's exchange.

  python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 1 tools/bench_extra/a2a_4gb.py
"""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
ok = True
def first_bad(a, b, step=1 << 28):
    total, first = 0, -1
    for lo in range(0, a.numel(), step):
        ne = a[lo:lo + step] != b[lo:lo + step]
        c = int(ne.sum())
        if c and first < 0:
            first = lo + int(ne.int().argmax())
        total += c
    return total, first


G = 1 << 30
for dtype, nbytes in ((torch.int32, G), (torch.int32, 2 * G - 4096), (torch.int32, 2 * G), (torch.int32, 2 * G + 4096), (torch.int32, 3 * G), (torch.int32, 4 * G - 4096),
                      (torch.int32, 4 * G), (torch.int32, 6 * G), (torch.uint8, 2 * G - 4096), (torch.uint8, 2 * G + 4096), (torch.uint8, 4 * G + 4096), (torch.int64, 4 * G + 4096)):
    esz = torch.empty(0, dtype=dtype).element_size()
    n = nbytes // esz
    src = torch.empty(n, dtype=dtype, device=dev)
    for lo in range(0, n, 1 << 28):                      # a pattern without a zero, made in bounded pieces
        hi = min(n, lo + (1 << 28))
        src[lo:hi] = (torch.arange(lo, hi, device=dev, dtype=torch.int64) * 2654435761 % 250 + 1).to(dtype)
    dst = torch.zeros(n, dtype=dtype, device=dev)
    dist.all_to_all_single(dst, src, output_split_sizes=[n], input_split_sizes=[n])
    torch.cuda.synchronize()
    bad, first = first_bad(dst, src)
    print(f"{str(dtype):12s} {nbytes / G:8.4f} GiB ({n} elements): mismatching elements {bad}" + (f", first at element {first} = byte {first * esz} ({first * esz / G:.4f} GiB)" if bad else ""), flush=True)
    ok = ok and bad == 0
    del src, dst
    torch.cuda.empty_cache()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
