#!/usr/bin/env python3
"""Practical HBM rates of plain streaming kernels on this MI355X, for comparison with k_corr_lookup's 478 MB per launch
(266 MB read + 212 MB written): a pure read (sum), a copy (1 read : 1 write) and an add (2 reads : 1 write) of the same total."""
import torch
dev = torch.device('cuda:0')


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]


for total_mb in (478, 1912):
    n3 = int(total_mb * 1e6 / 3 / 4)
    a = torch.randn(n3, device=dev); b = torch.randn(n3, device=dev); c = torch.empty(n3, device=dev)
    us = t(lambda: torch.add(a, b, out=c))
    print(f'add   (2R:1W) {3 * n3 * 4 / 1e6:7.0f} MB: {us:7.1f} us = {3 * n3 * 4 / us / 1e6:5.2f} TB/s')
    n2 = int(total_mb * 1e6 / 2 / 4)
    a = torch.randn(n2, device=dev); c = torch.empty(n2, device=dev)
    us = t(lambda: c.copy_(a))
    print(f'copy  (1R:1W) {2 * n2 * 4 / 1e6:7.0f} MB: {us:7.1f} us = {2 * n2 * 4 / us / 1e6:5.2f} TB/s')
    n1 = int(total_mb * 1e6 / 4)
    a = torch.randn(n1, device=dev)
    us = t(lambda: a.sum())
    print(f'sum   (read)  {n1 * 4 / 1e6:7.0f} MB: {us:7.1f} us = {n1 * 4 / us / 1e6:5.2f} TB/s  (two-stage reduction kernel)')
    c = torch.empty(n1, device=dev)
    us = t(lambda: c.fill_(1.0))
    print(f'fill  (write) {n1 * 4 / 1e6:7.0f} MB: {us:7.1f} us = {n1 * 4 / us / 1e6:5.2f} TB/s')
