"""usage (GPU box): python scripts/copy_sources.py [workload-model: s2anet|orcnn] -- which Python lines issue the layout /
dtype copies (aten::copy_, aten::contiguous, aten::clone, aten::fill_, aten::zero_, aten::add) of one train step: torch
profiler with stacks, device time summed per (op, first frame inside jdet_amd)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jdet_amd.models  # noqa: E402,F401
from jdet_amd.config.named import ORCNN_CFG, S2ANET_CFG  # noqa: E402
from jdet_amd.runner import Runner, synthetic_batch  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "s2anet"
dev = torch.device("cuda", 0)
r = Runner({"s2anet": S2ANET_CFG, "orcnn": ORCNN_CFG}[name], device=dev)
images, targets = synthetic_batch(2, 1024, dev, seed=1, num_gts=64)
for _ in range(4):
    r.train_step(images, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    r.train_step(images, targets)
    torch.cuda.synchronize()
WATCH = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::clone", "aten::cat", "aten::mul",
         "aten::mul_", "aten::sum", "aten::_foreach_norm", "aten::threshold_backward")
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.name not in WATCH or ev.device_time_total <= 0:
        continue
    frame = "?"
    for fr in ev.stack or []:
        if "jdet_amd" in fr and "runner.py" not in fr:
            frame = fr.split("jdet_amd/")[-1]
            break
    else:
        for fr in ev.stack or []:
            if "torch/" in fr and ("autograd" in fr or "optim" in fr or "nn/" in fr):
                frame = fr.split("torch/")[-1][:70]
                break
    shape = str(ev.input_shapes[0])[:40] if ev.input_shapes else ""
    k = (ev.name, frame[:90], shape)
    agg[k][0] += ev.device_time_total
    agg[k][1] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = collections.defaultdict(float)
for (n, _, _), (t, c) in rows:
    tot[n] += t
print("device us per step by op:", {k: round(v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
for (n, f, s), (t, c) in rows[:60]:
    print("%8.1f us %4d x  %-24s %-42s %s" % (t, c, n, s, f))
