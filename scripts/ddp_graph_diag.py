"""Two ranks on one device over gloo, HIP-graph mode (the set-up of tests/test_gpu_ddp_detectors.py::
test_two_ranks_graph_mode[orcnn]) with a probe at every hand-over of the step:

    own     norm of the rank's own flat gradient after its g1 replay (before the all-reduce)
    h_sum   checksum of the host tensor after the gloo all-reduce        (equal on both ranks by construction)
    h_dev   checksum of the device buffer after copy-back + division    (what g2 reads)
    h_par   checksum of all parameters after the step                   (what the test compares)

    python scripts/ddp_graph_diag.py <orcnn|s2anet> <steps> [runs]

One line per step with both ranks' values; the first step where the ranks disagree names the stage: h_sum differs ->
the collective; h_dev differs -> the copy-back; only h_par differs -> the captured update (g2); `own` far from its
neighbours -> a garbage gradient out of g1 (its worst parameter segments are printed by the rank that produced it)."""
import hashlib
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE = 256


def _hash(t):
    return hashlib.md5(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:10]


def worker(rank, world, port, name, steps):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      JDET_TRAIN_GRAPH_MULTI="1")     # the Runner refuses multi-rank graph steps otherwise
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jdet_amd.models  # noqa: F401
    from jdet_amd.config.named import ORCNN_CFG, S2ANET_CFG
    from jdet_amd.runner import Runner, synthetic_batch
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    r = Runner({"orcnn": ORCNN_CFG, "s2anet": S2ANET_CFG}[name], device=dev, conv_autotune=False, graph=True)
    segs, off = [], 0
    for n, p in r.model.named_parameters():
        if p.requires_grad:
            segs.append((n, off, off + p.numel()))
            off += p.numel()
    probe = {}
    owns = []

    def allreduce(flat):
        own = float(flat.norm())
        probe["own"] = own
        if owns and (own > 20 * max(owns) or own != own):
            d = sorted(((float(flat[a:b].norm()), n) for n, a, b in segs), reverse=True)[:6]
            print("[rank %d] GARBAGE own gradient %.4e (history max %.4e): %s" % (rank, own, max(owns), d), flush=True)
        owns.append(own)
        host = flat.cpu()
        dist.all_reduce(host)
        probe["h_sum"] = _hash(host)
        flat.copy_(host)
        flat.div_(world)
        probe["h_dev"] = _hash(flat)

    r._allreduce_flat = allreduce
    first_bad = None
    for step in range(steps):
        torch.manual_seed(9000 + 10 * step + rank)
        images, targets = synthetic_batch(1, SIZE, dev, seed=500 + 10 * (step % 3) + rank, num_gts=12)
        probe.clear()
        loss, _ = r.train_step(images, targets)
        flat = torch.cat([p.detach().reshape(-1) for p in r.model.parameters()])
        probe["h_par"] = _hash(flat)
        probe["loss"] = float(loss)
        probe["|par|"] = float(flat.norm())
        both = [None, None]
        dist.all_gather_object(both, dict(probe))
        if rank == 0:
            same = {k: both[0].get(k) == both[1].get(k) for k in ("h_sum", "h_dev", "h_par")}
            flag = "" if all(same.values()) else "   <-- RANKS DISAGREE: %s" % [k for k, v in same.items() if not v]
            print("step %d  own %.4e / %.4e  loss %.4f / %.4f  |par| %.6e / %.6e  %s%s"
                  % (step, both[0].get("own", 0.0), both[1].get("own", 0.0), both[0]["loss"], both[1]["loss"],
                     both[0]["|par|"], both[1]["|par|"], " ".join("%s=%s" % (k, both[0].get(k)) for k in ("h_sum", "h_dev", "h_par")),
                     flag), flush=True)
            if flag and first_bad is None:
                first_bad = step
    if rank == 0:
        print("RESULT: %s" % ("clean" if first_bad is None else "first disagreement at step %d" % first_bad), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    name, steps = sys.argv[1], int(sys.argv[2])
    runs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    for i in range(runs):
        port = 26000 + (os.getpid() + 13 * i) % 3000
        print("== run %d" % i, flush=True)
        mp.spawn(worker, args=(2, port, name, steps), nprocs=2, join=True)
