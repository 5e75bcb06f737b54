"""GPU: the graph-safe kernels (csrc/graph_safe.hip) and the property they exist for -- a captured train step holds NO
memset node (memset nodes do not reliably re-execute on replay on this stack: profiles/r05_graph_notes.md).  The node
kinds of the captured hipGraph are read through the runtime's graph API."""
import collections
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_zero_fill_and_sum_squares(dev):
    from jdet_amd import _lib as L
    g = torch.Generator(device="cpu").manual_seed(0)
    for n in (1, 3, 4, 1000, 4099, 38_000_000):
        x = torch.randn((n,), generator=g).to(dev)
        got = float(L.norm2(x))
        ref = float(torch.linalg.vector_norm(x.double()))
        assert abs(got - ref) <= 1e-6 * max(1.0, ref), (n, got, ref)
        assert float(L.norm2(x)) == got                                   # fixed summation order: same bits
    t = torch.full((2, 64, 33, 17), 3.0, device=dev).contiguous(memory_format=torch.channels_last)
    assert float(L.zero_(t).abs().sum()) == 0.0
    u = torch.full((1023,), 7, dtype=torch.int32, device=dev)
    assert int(L.zero_(u).abs().sum()) == 0


def node_kinds(graph):
    """Counter of node kinds + [(bytes, value)] of the memset nodes of a torch CUDAGraph created with keep_graph=True"""
    hip = ctypes.CDLL("libamdhip64.so")
    kind = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}

    class MemsetParams(ctypes.Structure):
        _fields_ = [("dst", ctypes.c_void_p), ("elementSize", ctypes.c_uint), ("height", ctypes.c_size_t),
                    ("pitch", ctypes.c_size_t), ("value", ctypes.c_uint), ("width", ctypes.c_size_t)]
    raw = ctypes.c_void_p(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) == 0
    nodes = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) == 0
    kinds, memsets = collections.Counter(), []
    for nd in nodes:
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
        kinds[kind.get(t.value, str(t.value))] += 1
        if t.value == 2:
            mp = MemsetParams()
            if hip.hipGraphMemsetNodeGetParams(ctypes.c_void_p(nd), ctypes.byref(mp)) == 0:
                memsets.append((mp.width * max(1, mp.height) * mp.elementSize, mp.value))
    return kinds, memsets


@pytest.mark.parametrize("which", ["s2anet", "orcnn"])
def test_captured_train_step_holds_no_memset_node(dev, which, monkeypatch):
    import jdet_amd.models  # noqa: F401
    from jdet_amd.config.named import ORCNN_CFG, S2ANET_CFG
    from jdet_amd.runner import Runner, synthetic_batch
    from jdet_amd import _lib as L
    graphs = []
    orig = L.new_graph

    def factory():
        g = orig()
        graphs.append(g)
        return g
    monkeypatch.setattr(L, "new_graph", factory)
    torch.manual_seed(0)
    runner = Runner(S2ANET_CFG if which == "s2anet" else ORCNN_CFG, device=dev, graph=True)
    images, targets = synthetic_batch(2, 512, dev, seed=3)
    images = images.contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        runner.train_step(images, targets)
    torch.cuda.synchronize()
    assert graphs
    for g in graphs:
        kinds, memsets = node_kinds(g)              # AFTER Runner's hardening pass (and the replays)
        assert kinds["kernel"] > 500
        assert kinds["memset"] == 0, memsets


def test_memset_nodes_become_kernels_and_the_graph_still_computes(dev):
    """a captured graph with framework reductions (multi-workgroup sum / norm: semaphores cleared by memset nodes) and
    plain memsets of every element size: after jdet_graph_replace_memset_nodes no memset node is left and 20 replays
    give the eager results, with the buffers dirtied between replays"""
    from jdet_amd import _lib as L
    big = torch.randn(1024, 1024, device=dev)
    buf8 = torch.full((1000,), 7, dtype=torch.uint8, device=dev)
    buf32 = torch.full((4097,), 9.0, device=dev)
    out = {}

    def work():
        out["s"] = big.sum(0)
        out["n"] = torch.linalg.vector_norm(big)
        out["m"] = big.max(0).values
        buf8.zero_()
        buf32.zero_()
        out["z"] = buf8.sum() + buf32.sum()
    for _ in range(2):
        work()
    torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in out.items()}
    g = L.new_graph()
    with torch.cuda.graph(g):
        work()
    before, _ = node_kinds(g)
    replaced = L.harden_graph(g)
    after, _ = node_kinds(g)
    assert before["memset"] >= 3 and replaced == before["memset"] and after["memset"] == 0
    assert after["kernel"] == before["kernel"] + replaced
    for _ in range(20):
        buf8.fill_(5)
        buf32.fill_(3.0)
        g.replay()
        torch.cuda.synchronize()
        for k in ref:
            torch.testing.assert_close(out[k], ref[k], rtol=1e-6, atol=1e-6)
