"""The bottleneck convolution family (csrc/conv_bn.hip) at the ResNet-50 shapes of a 2 x 1024^2 step, per layer and per
block, against the per-layer path it replaces (library convolution + frozen-BN pass):
  python scripts/conv_bn_timing.py [layers|tiles|blocks|wgrad ...]     (default: all)
layers: forward conv+bn(+res)+relu, own kernel vs library conv + jdet_frozen_bn_act_forward, and the MASK data gradient vs
        library dgrad + frozen-BN backward;  tiles: the forced tile shapes of the own kernel per layer;
blocks: one Bottleneck forward + backward, fused vs per-layer;  wgrad: own general weight gradient vs the library's."""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from jdet_amd import _lib as L  # noqa: E402
from jdet_amd.ops import conv_bn as CB  # noqa: E402
from jdet_amd.ops.frozen_bn import frozen_bn_act  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
# (name, N, H, W, Cin, Cout, R, stride, residual) -- the distinct conv layers of ResNet-50 at 2 x 1024^2
LAYERS = [
    ("l1.conv1", 2, 256, 256, 256, 64, 1, 1, False), ("l1.conv2", 2, 256, 256, 64, 64, 3, 1, False),
    ("l1.conv3", 2, 256, 256, 64, 256, 1, 1, True),
    ("l2.0.conv1", 2, 256, 256, 256, 128, 1, 1, False), ("l2.0.conv2", 2, 256, 256, 128, 128, 3, 2, False),
    ("l2.0.down", 2, 256, 256, 256, 512, 1, 2, False),
    ("l2.conv1", 2, 128, 128, 512, 128, 1, 1, False), ("l2.conv2", 2, 128, 128, 128, 128, 3, 1, False),
    ("l2.conv3", 2, 128, 128, 128, 512, 1, 1, True),
    ("l3.0.conv2", 2, 128, 128, 256, 256, 3, 2, False),
    ("l3.conv1", 2, 64, 64, 1024, 256, 1, 1, False), ("l3.conv2", 2, 64, 64, 256, 256, 3, 1, False),
    ("l3.conv3", 2, 64, 64, 256, 1024, 1, 1, True),
    ("l4.0.conv2", 2, 64, 64, 512, 512, 3, 2, False),
    ("l4.conv1", 2, 32, 32, 2048, 512, 1, 1, False), ("l4.conv2", 2, 32, 32, 512, 512, 3, 1, False),
    ("l4.conv3", 2, 32, 32, 512, 2048, 1, 1, True),
]


def timeit(fn, n=30):
    # (25 warm-up launches: with 5, the first variant timed for a layer ran 5-9 % slow -- clocks and caches still settling --
    #  which read as a win for whatever came later in a sweep: profiles/r06_conv_prefetch.md)
    for _ in range(25):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def make_bn(C):
    bn = torch.nn.BatchNorm2d(C).to(dev).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.running_var.uniform_(0.5, 1.5)
        bn.running_mean.normal_(0, 0.2)
    return bn


def layers():
    print("# forward: own conv+bn[+res]+relu vs library conv + frozen-BN pass | backward: own MASK dgrad (stride 1) vs library "
          "dgrad + frozen-BN backward of the layer below; us")
    tf = tl = 0.0
    for name, N, H, W, Ci, Co, R, s, res in LAYERS:
        x = torch.randn(N, H, W, Ci, device=dev)
        w = torch.randn(Co, R, R, Ci, device=dev) / (R * Ci ** 0.5)
        bn = make_bn(Co)
        Ho, Wo = CB.out_size(H, R, s), CB.out_size(W, R, s)
        r = torch.randn(N, Ho, Wo, Co, device=dev) if res else None
        own = timeit(lambda: CB.conv_bn_nhwc(x, w, s, bn, r, True))
        xc, wc = x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2)
        rc = r.permute(0, 3, 1, 2) if res else None
        conv_only = timeit(lambda: F.conv2d(xc, wc, None, s, R // 2))
        lib = timeit(lambda: frozen_bn_act(F.conv2d(xc, wc, None, s, R // 2), bn, residual=rc))
        flops = 2.0 * N * Ho * Wo * Co * R * R * Ci
        nbytes = 4.0 * (N * H * W * Ci + N * Ho * Wo * Co * (2 if res else 1))
        line = "%-11s fwd own %7.1f | lib conv %7.1f conv+bn %7.1f | %5.1f TF/s %5.2f TB/s" % (
            name, own, conv_only, lib, flops / own / 1e6, nbytes / own / 1e6)
        tf += own
        tl += lib
        if s == 1:
            # data gradient w.r.t. this layer's input, masked / scaled for the layer below (Cin channels)
            gy = torch.randn(N, Ho, Wo, Co, device=dev)
            wt = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
            bnb = make_bn(Ci)
            act = torch.relu(torch.randn(N, H, W, Ci, device=dev))
            cb = torch.randn(N, H, W, Ci, device=dev)
            ownb = timeit(lambda: CB.conv_bn_nhwc(gy, wt, 1, bnb, mode=L.EPI_MASK, act=act, want_sums=True))
            gyc = gy.permute(0, 3, 1, 2)

            def libb():
                gx = torch.ops.aten.convolution_backward(gyc, xc, wc, None, [1, 1], [R // 2, R // 2], [1, 1], False, [0, 0],
                                                         1, [True, False, False])[0]
                P = N * H * W
                gxx = torch.empty_like(gx)
                gw_, gb_ = torch.empty(Ci, device=dev), torch.empty(Ci, device=dev)
                wsb = L.lib().jdet_frozen_bn_act_backward_workspace(P, Ci)
                ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
                L.check(L.lib().jdet_frozen_bn_act_backward(L.ptr(gx), L.ptr(act), L.ptr(cb), P, Ci, L.ptr(bnb.weight),
                                                            L.ptr(bnb.bias), L.ptr(bnb.running_mean), L.ptr(bnb.running_var),
                                                            bnb.eps, 1, L.ptr(gxx), None, L.ptr(gw_), L.ptr(gb_), L.ptr(ws),
                                                            wsb, L.stream_ptr(gx)), "bn bwd")
            line += " || dgrad+mask own %7.1f lib %7.1f" % (ownb, timeit(libb))
        print(line, flush=True)
    print("sum forward: own %.0f us, library conv + bn %.0f us" % (tf, tl))


def tiles():
    print("# forced tiles of the own forward kernel per layer (0 = the heuristic); us")
    for name, N, H, W, Ci, Co, R, s, res in LAYERS:
        x = torch.randn(N, H, W, Ci, device=dev)
        w = torch.randn(Co, R, R, Ci, device=dev) / (R * Ci ** 0.5)
        bn = make_bn(Co)
        out = []
        Ho, Wo = CB.out_size(H, R, s), CB.out_size(W, R, s)
        r = torch.randn(N, Ho, Wo, Co, device=dev) if res else None
        for t in (0, 64, 65, 66, 67, 128, 130):       # +1: 16-deep K steps, +2: no intra-workgroup K split
            out.append("%d: %6.1f" % (t, timeit(lambda: CB.conv_bn_nhwc(x, w, s, bn, r, True, tile=t), 20)))
        print("%-11s %s" % (name, "  ".join(out)), flush=True)


def wgrad():
    print("# weight gradient: own general kernel (auto split | forced) vs the library's; us")
    for name, N, H, W, Ci, Co, R, s, res in LAYERS:
        x = torch.randn(N, H, W, Ci, device=dev)
        Ho, Wo = CB.out_size(H, R, s), CB.out_size(W, R, s)
        gy = torch.randn(N, Ho, Wo, Co, device=dev)
        out = torch.zeros(Co, R, R, Ci, device=dev)
        xc, gyc = x.permute(0, 3, 1, 2), gy.permute(0, 3, 1, 2)
        wc = torch.randn(Co, R, R, Ci, device=dev).permute(0, 3, 1, 2)
        lib = timeit(lambda: torch.ops.aten.convolution_backward(gyc, xc, wc, None, [s, s], [R // 2, R // 2], [1, 1], False,
                                                                 [0, 0], 1, [False, True, False]))
        S = 1 << 17            # 64 x 64 tiles
        ts = ["%s: %6.1f" % (n, timeit(lambda: CB.conv_wgrad_nhwc(x, gy, R, s, out, k), 20))
              for n, k in (("auto", 0), ("s4", S | 4), ("s8", S | 8), ("s16", S | 16), ("s32", S | 32), ("s64", S | 64),
                           ("s128", S | 128), ("s256", S | 256), ("big", 1 << 18))]
        print("%-11s lib %7.1f | own %s" % (name, lib, "  ".join(ts)), flush=True)


def blocks():
    from jdet_amd.models.backbones.resnet import Bottleneck, conv1x1
    print("# one Bottleneck forward + backward (all gradients), fused vs per-layer path; us")
    for name, inpl, pl, st, ds, H in (("layer2.1", 512, 128, 1, False, 128), ("layer3.1", 1024, 256, 1, False, 64),
                                      ("layer4.1", 2048, 512, 1, False, 32), ("layer2.0", 256, 128, 2, True, 256),
                                      ("layer3.0", 512, 256, 2, True, 128), ("layer4.0", 1024, 512, 2, True, 64)):
        d = torch.nn.Sequential(conv1x1(inpl, pl * 4, st), torch.nn.BatchNorm2d(pl * 4)) if ds else None
        blk = Bottleneck(inpl, pl, st, d).to(dev).eval()
        for p in blk.parameters():
            if p.dim() == 4:
                p.data = p.data.contiguous(memory_format=torch.channels_last)
        x = torch.randn(2, inpl, H, H, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        gy = None
        res = []
        for fused in (True, False):
            CB.ENABLED = fused

            def step():
                nonlocal gy
                y = blk(x)
                if gy is None:
                    gy = torch.randn_like(y)
                y.backward(gy)
                blk.zero_grad(set_to_none=True)
                x.grad = None
            res.append(timeit(step, 20))
            t0 = time.perf_counter()
            for _ in range(20):
                step()
            host = (time.perf_counter() - t0) / 20 * 1e6
            torch.cuda.synchronize()
            res.append(host)
        CB.ENABLED = True
        print("%-9s fused %7.1f (host issue %6.0f) | per-layer %7.1f (host issue %6.0f)" % (name, *res), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["layers", "tiles", "wgrad", "blocks"]
    for wname in what:
        globals()[wname]()
