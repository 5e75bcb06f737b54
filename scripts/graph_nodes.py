#!/usr/bin/env python3
"""Which node types does a captured train step hold?  (round 6, VERDICT r5 item 7: memset nodes do not reliably re-execute
on replay on this stack -- every one left in a captured region is a latent garbage source.)
    python scripts/graph_nodes.py s2anet|orcnn|roitrans [size]
Captures the step through Runner(graph=True) with the graph's debug mode on, dumps it as DOT and counts the node kinds;
memset nodes are listed with their byte counts."""
import collections
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jdet_amd.models  # noqa: E402,F401
from jdet_amd.config.named import ORCNN_CFG, S2ANET_CFG, roitrans_train_cfg  # noqa: E402
from jdet_amd.runner import Runner, synthetic_batch  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "s2anet"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = {"s2anet": lambda: S2ANET_CFG, "orcnn": lambda: ORCNN_CFG, "roitrans": lambda: roitrans_train_cfg("Resnet50")}[which]()
dev = torch.device("cuda:0")
graphs = []
_orig = torch.cuda.CUDAGraph


def _factory(*a, **k):
    g = _orig(keep_graph=True)          # keeps the hipGraph_t: raw_cuda_graph()
    graphs.append(g)
    return g


torch.cuda.CUDAGraph = _factory
torch.manual_seed(0)
runner = Runner(cfg, device=dev, graph=True)
images, targets = synthetic_batch(2, size, dev, seed=3)
images = images.contiguous(memory_format=torch.channels_last)
for _ in range(3):
    runner.train_step(images, targets)
torch.cuda.synchronize()

import ctypes  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
KIND = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record",
        8: "ext_sem_signal", 9: "ext_sem_wait", 10: "mem_alloc", 11: "mem_free", 12: "memcpy_from_symbol",
        13: "memcpy_to_symbol"}


class MemsetParams(ctypes.Structure):      # hipMemsetParams
    _fields_ = [("dst", ctypes.c_void_p), ("elementSize", ctypes.c_uint), ("height", ctypes.c_size_t),
                ("pitch", ctypes.c_size_t), ("value", ctypes.c_uint), ("width", ctypes.c_size_t)]


for i, g in enumerate(graphs):
    raw = ctypes.c_void_p(g.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) == 0
    nodes = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) == 0
    kinds = collections.Counter()
    memsets = []
    for nd in nodes:
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
        kinds[KIND.get(t.value, str(t.value))] += 1
        if t.value == 2:
            mp = MemsetParams()
            if hip.hipGraphMemsetNodeGetParams(ctypes.c_void_p(nd), ctypes.byref(mp)) == 0:
                memsets.append((mp.width * max(1, mp.height) * mp.elementSize, mp.value))
    # who consumes a memset node's output: the kernels that depend on it
    class Dim3(ctypes.Structure):
        _fields_ = [("x", ctypes.c_uint), ("y", ctypes.c_uint), ("z", ctypes.c_uint)]

    class KernelParams(ctypes.Structure):      # hipKernelNodeParams
        _fields_ = [("blockDim", Dim3), ("extra", ctypes.c_void_p), ("func", ctypes.c_void_p), ("gridDim", Dim3),
                    ("kernelParams", ctypes.c_void_p), ("sharedMemBytes", ctypes.c_uint)]
    hip.hipKernelNameRefByPtr.restype = ctypes.c_char_p
    hip.hipKernelNameRef.restype = ctypes.c_char_p
    for nd in nodes:
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
        if t.value != 2:
            continue
        mp = MemsetParams()
        hip.hipGraphMemsetNodeGetParams(ctypes.c_void_p(nd), ctypes.byref(mp))
        nd_n = ctypes.c_size_t(0)
        hip.hipGraphNodeGetDependentNodes(ctypes.c_void_p(nd), None, ctypes.byref(nd_n))
        deps = (ctypes.c_void_p * max(1, nd_n.value))()
        hip.hipGraphNodeGetDependentNodes(ctypes.c_void_p(nd), deps, ctypes.byref(nd_n))
        names = []
        for d in list(deps)[:nd_n.value]:
            dt = ctypes.c_int(-1)
            hip.hipGraphNodeGetType(ctypes.c_void_p(d), ctypes.byref(dt))
            if dt.value == 0:
                kp = KernelParams()
                if hip.hipGraphKernelNodeGetParams(ctypes.c_void_p(d), ctypes.byref(kp)) == 0:
                    nm = hip.hipKernelNameRefByPtr(ctypes.c_void_p(kp.func), None) or hip.hipKernelNameRef(ctypes.c_void_p(kp.func))
                    names.append("%s grid %d block %d" % ((nm or b"?").decode()[:150], kp.gridDim.x, kp.blockDim.x))
            else:
                names.append(KIND.get(dt.value, str(dt.value)))
        print("   memset of %d bytes -> %s" % (mp.width * max(1, mp.height) * mp.elementSize, names))
    print("graph %d of %s: %d nodes %s" % (i, which, n.value, dict(kinds)))
    by = collections.Counter(memsets)
    for (nbytes, val), c in sorted(by.items()):
        print("   memset nodes: %4d x %10d bytes, value %d" % (c, nbytes, val))
