#!/usr/bin/env python3
"""Who launches the small framework kernels of a train step?  (GPU box.)

Runs a few eager steps of a workload under torch.profiler and prints, for every device kernel whose name matches one
of the given substrings, the chain of host-side operator names above the launch (autograd node, aten op) with the
launch count and the device time per step.

usage: glue_attrib.py [workload] [substring ...]      (default: s2anet_train  direct_copy CUDAFunctor_add SubTensorOp)
"""
import argparse
import sys
from collections import defaultdict

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "s2anet_train"
    pats = sys.argv[2:] or ["direct_copy", "CUDAFunctor_add", "SubTensorOp", "AUnaryFunctor", "BinaryFunctor",
                            "FillFunctor", "CatArray"]
    a = argparse.Namespace(workload=wl, batch=4 if wl == "roitrans_train" else 2,
                           size=1024, amp="none", steps=3, warmup=8)
    dev = torch.device("cuda:0")
    step, keep = bench.make_train(a, 0, dev)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    steps = 3
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.kernels:
            continue
        chain, e = [], ev
        while e is not None:
            chain.append(e.name)
            e = e.cpu_parent
        for k in ev.kernels:
            if any(p in k.name for p in pats):
                key = (next(p for p in pats if p in k.name), " <- ".join(c[:60] for c in chain[:5]))
                agg[key][0] += 1
                agg[key][1] += k.duration
    for (pat, chain), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print("%6.1f /step %8.1f us/step  %-18s %s" % (n / steps, us / steps, pat, chain))


if __name__ == "__main__":
    main()
