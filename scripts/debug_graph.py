"""bisect the HIP-graph train step: python scripts/debug_graph.py <mode> <size>
modes: fwd | fwdbwd | full | head (FPN features fixed, head+loss fwd/bwd only) | backbone"""
import sys
import torch
sys.path.insert(0, ".")
from jdet_amd.config.named import S2ANET_CFG
from jdet_amd.runner import Runner, synthetic_batch
from jdet_amd.utils.general import parse_losses

mode, size = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
torch.manual_seed(0)
r = Runner(S2ANET_CFG, device=dev, conv_autotune=False, graph=False)
images, targets = synthetic_batch(2, size, dev, seed=3, num_gts=16)
images = images.contiguous(memory_format=torch.channels_last)
for _ in range(2):
    r.train_step(images, targets)
torch.cuda.synchronize()
print("eager ok", flush=True)
m = r.model
m.train()
params = [p for p in m.parameters() if p.requires_grad]
for p in params:
    p.grad = None

if mode in ("head", "backbone"):
    with torch.no_grad():
        feats = m.neck(m.backbone(images))
    feats = [f.detach().clone().requires_grad_(True) for f in feats]


def body():
    if mode == "backbone":
        out = m.neck(m.backbone(images))
        loss = sum(o.mean() for o in out)
        loss.backward()
        return loss.detach()
    if mode == "head":
        losses = m.bbox_head(feats, targets)
    else:
        losses = m(images, targets)
    total, _ = parse_losses(losses)
    if mode != "fwd":
        total.backward()
    return total.detach()


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("side warmup ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
torch.cuda.synchronize()
print("capture ok", flush=True)
for i in range(4):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, float(out), flush=True)
print("DONE", mode)
