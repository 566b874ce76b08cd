#!/usr/bin/env python3
"""Where the time of a large fresh device allocation goes (the eight 50k cost matrices are 160 GB): the allocation call, the
first write, a second write; one piece or eight, from one thread or eight.  Tools only."""
import sys
import threading
import time

import torch

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 160.0
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
torch.cuda.synchronize()


def t(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print("%-60s %8.1f ms" % (label, (time.perf_counter() - t0) * 1e3), flush=True)
    return out


n = int(gb * 1e9 / 8)
x = t("torch.empty, %.0f GB in one piece" % gb, lambda: torch.empty(n, dtype=torch.float64, device=dev))
t("first write (fill_)", lambda: x.fill_(1.0))
t("second write", lambda: x.fill_(2.0))
del x
t("empty_cache (free)", torch.cuda.empty_cache)
xs = t("8 pieces, one thread", lambda: [torch.empty(n // 8, dtype=torch.float64, device=dev) for _ in range(8)])
t("first write of the 8 pieces", lambda: [y.fill_(1.0) for y in xs])
del xs
torch.cuda.empty_cache()
out = [None] * 8


def one(i):
    out[i] = torch.empty(n // 8, dtype=torch.float64, device=dev)


def par():
    th = [threading.Thread(target=one, args=(i,)) for i in range(8)]
    [x.start() for x in th]
    [x.join() for x in th]


t("8 pieces, eight threads", par)
t("first write of the 8 pieces", lambda: [y.fill_(1.0) for y in out])
