"""Root-causing the Oriented R-CNN HIP-graph divergence: ONE process captures the train step the
way a rank of a two-rank job does (g1 = forward + losses + backward into the flat gradient buffer, no update), then
replays g1 many times FROM IDENTICAL STATE (same inputs, same generator seed, parameters untouched) and compares the
flat gradient of every replay with the first one.

    python scripts/graph_replay_diag.py <orcnn|s2anet> <replays> [tag]

Replays legitimately differ in the last bits (the library's split-K weight gradients and the sorted gather sum in
arrival order; proposal NMS / sampling amplify that to ~1e-1 of the update norm for Oriented R-CNN) -- what must never
happen is a GARBAGE replay: relative distance >= 1, a non-finite value, or a norm far from the reference's.  Printed
per replay only when it is suspicious, with the parameters whose segment went wrong (largest first) -- the map from
"which gradient is garbage" to "which kernel wrote it".  Run two of these at once (scripts/gpu_r4_graph.sh) to put
the replays under the time slicing of two processes on one device, which is what the failing test does.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name, replays = sys.argv[1], int(sys.argv[2])
tag = sys.argv[3] if len(sys.argv) > 3 else "solo"
SIZE = int(os.environ.get("JDET_DIAG_SIZE", "256"))

import jdet_amd.models  # noqa: E402,F401
from jdet_amd.config.named import ORCNN_CFG, S2ANET_CFG  # noqa: E402
from jdet_amd.runner import Runner, synthetic_batch  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
r = Runner({"orcnn": ORCNN_CFG, "s2anet": S2ANET_CFG}[name], device=dev, conv_autotune=False, graph=True, ddp=False)
images, targets = synthetic_batch(1, SIZE, dev, seed=500, num_gts=12)
images = images.contiguous(memory_format=torch.channels_last)
for step in range(2):                      # the eager warm-up steps of graph mode
    torch.manual_seed(9000 + step)
    r.train_step(images, targets)
r.world_size = 2                           # capture like a rank of a two-rank job: g1 without the update, g2 = update
r.model.train()
st = r._capture(images, targets)
r.world_size = 1
torch.cuda.synchronize()
flat = st["flat"]
names, off, segs = [], 0, []
for n, p in r.model.named_parameters():
    if p.requires_grad:
        segs.append((n, off, off + p.numel()))
        off += p.numel()


def replay(seed):
    torch.manual_seed(seed)                # the samplers' random keys: graph-safe generator state, re-armed per replay
    st["g1"].replay()
    torch.cuda.synchronize()
    return flat.clone(), float(st["out"][0])


ref, loss0 = replay(777)
rn = float(ref.norm())
print("[%s] %s size %d: reference gradient norm %.6e loss %.6f finite %s" % (tag, name, SIZE, rn, loss0,
                                                                            bool(torch.isfinite(ref).all())), flush=True)
worst, bad, t0 = 0.0, 0, time.time()
for i in range(replays):
    g, loss = replay(777)
    rel = float((g - ref).norm()) / rn
    finite = bool(torch.isfinite(g).all())
    worst = max(worst, rel if finite else float("inf"))
    if not finite or rel >= 0.5 or abs(float(g.norm()) / rn - 1.0) > 0.5:
        bad += 1
        d = []
        for n, a, b in segs:
            sr, sg = ref[a:b], g[a:b]
            e = float((sg - sr).norm())
            if not torch.isfinite(sg).all():
                e = float("inf")
            d.append((e, n, float(sr.norm()), float(sg.norm())))
        d.sort(reverse=True)
        print("[%s] replay %d GARBAGE: rel %.3e |g| %.4e loss %.6f finite %s; worst segments: %s"
              % (tag, i, rel, float(g.norm()), loss, finite,
                 "; ".join("%s d=%.3e ref=%.3e got=%.3e" % (n, e, a, b) for e, n, a, b in d[:6])), flush=True)
print("[%s] %d replays in %.1f s: %d garbage, worst relative distance %.3e" % (tag, replays, time.time() - t0, bad, worst),
      flush=True)
