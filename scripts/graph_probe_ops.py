#!/usr/bin/env python3
"""Which framework ops put memset nodes into a captured graph?  (round 6)  Each candidate is captured alone; the node
kinds of its graph are printed (tests/test_gpu_graph_safe.py::node_kinds)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_graph_safe import node_kinds  # noqa: E402

dev = torch.device("cuda:0")
x = torch.rand(261888, device=dev)
m = torch.rand(2000, device=dev)
big = torch.randn(1024, 1024, device=dev)
w = torch.randn(1024, 12544, device=dev)
a = torch.randn(1024, 12544, device=dev)
idx = torch.randint(0, 261888, (5000,), device=dev)
cases = {
    "topk 128 of 261888": lambda: torch.topk(x, 128, largest=False),
    "topk 2000 of 261888": lambda: torch.topk(x, 2000),
    "topk 512 of 2000": lambda: torch.topk(m, 512),
    "sort 261888": lambda: torch.sort(x),
    "cumsum 261888": lambda: torch.cumsum(x, 0),
    "sum (1024,1024) dim 0": lambda: big.sum(0),
    "vector_norm 1M": lambda: torch.linalg.vector_norm(big),
    "linear 1024x12544": lambda: torch.nn.functional.linear(a, w),
    "zeros 32 MiB": lambda: torch.zeros(8 * 1024 * 1024, device=dev),
    "zeros 4 KiB": lambda: torch.zeros(1024, device=dev),
    "index_put": lambda: x.index_put((idx,), torch.ones(5000, device=dev)),
    "max dim": lambda: big.max(0),
    "argsort 2000": lambda: torch.argsort(m),
}
for name, fn in cases.items():
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        fn()
    kinds, memsets = node_kinds(g)
    print("%-26s %s %s" % (name, dict(kinds), memsets))
