import sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from jdet_amd.models.backbones.resnet import Resnet50
from jdet_amd.ops import conv_bn as CB
dev = torch.device("cuda:0")
torch.manual_seed(11)
m = Resnet50(return_stages=["layer1", "layer2", "layer3", "layer4"], frozen_stages=1, norm_eval=True).to(dev).train()
for p in m.parameters():
    if p.dim() == 4 and p.shape[2] * p.shape[3] > 1:
        p.data = p.data.contiguous(memory_format=torch.channels_last)
x = torch.randn(2, 3, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
gys = None
res = {}
for name, fused, own in (("own", True, True), ("lib", True, False), ("own2", True, True), ("per", False, False)):
    CB.ENABLED, CB.OWN_WGRAD = fused, own
    m.zero_grad(set_to_none=True)
    outs = m(x)
    if gys is None:
        gys = [torch.randn_like(o) for o in outs[1:]]
    torch.autograd.backward(outs[1:], gys)
    res[name] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
CB.ENABLED, CB.OWN_WGRAD = True, True
for a, b in (("own", "lib"), ("own", "own2"), ("lib", "per"), ("own", "per")):
    d = sorted(((float((res[a][n] - res[b][n]).abs().max() / (res[b][n].abs().max() + 1e-6)), n) for n in res[a]), reverse=True)[:6]
    print(a, "vs", b, [(round(v, 6), n) for v, n in d])
# the shape in isolation
g = torch.Generator().manual_seed(1)
for (N, H, W, Ci, Co) in ((2, 16, 16, 512, 128), (2, 16, 16, 128, 512), (2, 8, 8, 1024, 256), (2, 4, 4, 2048, 512)):
    xx = torch.randn(N, H, W, Ci, generator=g).to(dev)
    gy = torch.randn(N, H, W, Co, generator=g).to(dev)
    out = torch.zeros(Co, 1, 1, Ci, device=dev)
    CB.conv_wgrad_nhwc(xx, gy, 1, 1, out)
    ref = (gy.reshape(-1, Co).double().t() @ xx.reshape(-1, Ci).double()).view(Co, 1, 1, Ci)
    print((N, H, W, Ci, Co), float((out.double() - ref).abs().max() / ref.abs().max()))
