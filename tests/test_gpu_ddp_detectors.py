"""GPU, two ranks on ONE device over gloo (CUDA tensors): the production Runner with the real detectors under
world_size 2 -- the reference's data-parallel semantics (runner.py:L117-155, optimizer.py:L26-36: every rank its own
tiles, gradients averaged over ranks before the update).  Asserted per config:
  * DDP eager: parameters bit-identical across ranks after every step, and equal to a single process that averages
    the two ranks' gradients by hand (the definition of the all-reduce) -- NOT to the loss of the concatenated batch:
    the detection losses are normalised per rank (sum over the rank's images of max(#pos, 1)), as in the reference;
  * HIP-graph steps under world_size 2 (un-refused in round 6: every captured graph is stripped of memset nodes before
    instantiation, csrc/graph_safe.hip): two eager warm-up steps, one capture, 20 replays -- replicas bit-identical after
    every step, parameters finite and moving; JDET_TRAIN_GRAPH_MULTI=0 still falls back to eager DDP.
RCCL with more than one rank needs more than one GPU; the driver's 8-GPU run covers it."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE, STEPS = 256, 3


def _cfg(name):
    from jdet_amd.config.named import ORCNN_CFG, S2ANET_CFG
    return {"s2anet": S2ANET_CFG, "orcnn": ORCNN_CFG}[name]


def _batch(step, rank, dev, size=SIZE):
    from jdet_amd.runner import synthetic_batch
    return synthetic_batch(1, size, dev, seed=500 + 10 * step + rank, num_gts=12)


def _seed_step(step, rank):
    torch.manual_seed(9000 + 10 * step + rank)       # the samplers' random keys (two-stage heads)


REPLAYS = 20


def _worker(rank, world, port, name, graph, out, size=SIZE):
    sys.path.insert(0, ROOT)
    import warnings
    # the stream-mismatch warning of the autograd engine (round 4's bench stderr) is an error here
    warnings.filterwarnings("error", message=".*AccumulateGrad node's stream.*")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if size != SIZE:      # the side-stream configuration (opt-in since round 6) is what that test is about
            os.environ["JDET_HEAD_STREAMS"] = "1"
        import jdet_amd.models  # noqa: F401
        from jdet_amd.runner import Runner
        dev = torch.device("cuda", 0)
        torch.manual_seed(1234)                       # identical replicas
        r = Runner(_cfg(name), device=dev, conv_autotune=False, graph=graph)
        assert r.world_size == world
        if graph:
            assert r.use_graph == (os.environ.get("JDET_TRAIN_GRAPH_MULTI", "1") != "0")
        history = [torch.cat([p.detach().reshape(-1) for p in r.model.parameters()])[::97].cpu()]
        nsteps = (3 + REPLAYS) if graph else STEPS    # graph mode: two eager warm-up steps, one capture, replays
        for step in range(nsteps):
            _seed_step(step, rank)
            images, targets = _batch(step % STEPS if graph else step, rank, dev, size)
            loss, _ = r.train_step(images, targets)
            assert torch.isfinite(loss).all()
            flat = torch.cat([p.detach().reshape(-1) for p in r.model.parameters()]).cpu()
            other = flat.clone()
            dist.broadcast(other, 0)                 # host tensors: the check itself uses no device collective
            if not torch.equal(other, flat):
                d = (other - flat).abs()
                raise AssertionError("replicas diverged at step %d (rank %d): %d elements differ, max %.3e, non-finite %d"
                                     "; |mine| %.4e |rank 0| %.4e; history of |mine|: %s"
                                     % (step, rank, int((d > 0).sum()), float(d.max()),
                                        int((~torch.isfinite(flat)).sum()), float(flat.norm()), float(other.norm()),
                                        [float(h.norm()) for h in history]))
            history.append(flat[::97].clone())
        if size != SIZE and name == "s2anet":
            from jdet_amd.models.roi_heads import s2anet_head
            assert s2anet_head.HEAD_STREAMS and len(s2anet_head._SIDE) == 1, "the packed levels did not take the side stream"
        if rank == 0:
            torch.save(history, out)
    finally:
        dist.destroy_process_group()


def _single_process_reference(name, dev, size=SIZE):
    """one process: per step the two ranks' gradients on identical parameters, averaged, then the optimizer's update"""
    import jdet_amd.models  # noqa: F401
    from jdet_amd.runner import Runner
    from jdet_amd.utils.general import parse_losses
    torch.manual_seed(1234)
    r = Runner(_cfg(name), device=dev, conv_autotune=False, ddp=False, graph=False)
    params = [p for p in r.model.parameters() if p.requires_grad]
    history = [torch.cat([p.detach().reshape(-1) for p in r.model.parameters()])[::97].cpu()]
    for step in range(STEPS):
        grads = []
        for rank in range(2):
            _seed_step(step, rank)
            images, targets = _batch(step, rank, dev, size)
            r.model.train()
            loss, _ = parse_losses(r.model(images.contiguous(memory_format=torch.channels_last), targets))
            r.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            grads.append([None if p.grad is None else p.grad.clone() for p in params])
        r.optimizer.zero_grad(set_to_none=True)
        for p, g0, g1 in zip(params, *grads):
            if g0 is None and g1 is None:
                continue
            z = torch.zeros_like(p)
            p.grad = ((z if g0 is None else g0) + (z if g1 is None else g1)) / 2
        r.optimizer.step(None)
        if r.scheduler is not None:
            r.scheduler.step(r.iter, r.epoch, by_epoch=True)
        r.iter += 1
        history.append(torch.cat([p.detach().reshape(-1) for p in r.model.parameters()])[::97].cpu())
    return history


@pytest.mark.parametrize("name", ["s2anet", "orcnn"])
def test_two_ranks_ddp_eager(dev, tmp_path, name):
    port = 23000 + os.getpid() % 2000 + (7 if name == "orcnn" else 0)
    out = str(tmp_path / "h.pt")
    mp.spawn(_worker, args=(2, port, name, False, out), nprocs=2, join=True)
    got = torch.load(out)
    ref = _single_process_reference(name, dev)
    assert torch.equal(got[0], ref[0])                 # identical initial replicas
    for step in range(1, STEPS + 1):
        # same arithmetic up to summation order (all-reduce, the library kernels' atomics).  The yardstick is the
        # run-to-run spread of the single-process computation itself, measured on the UPDATE of a step (a sample of
        # every 97th parameter) relative to its norm: S2ANet 1e-4 .. 2e-3; Oriented R-CNN 1.6e-2 at the first step and
        # 6e-2 .. 1e-1 later (proposal NMS, sampling and assignment turn last-bit differences into different RoIs).
        # Bounds a few times that -- a sum instead of a mean, or a missing all-reduce, is off by 50-100 % at step 1
        da, db = got[step] - got[step - 1], ref[step] - ref[step - 1]
        assert float(db.norm()) > 0
        first, later = (1e-2, 2e-2) if name == "s2anet" else (6e-2, 0.35)
        assert float((da - db).norm()) <= (first if step == 1 else later) * float(db.norm()), (name, step, float((da - db).norm() / db.norm()))


def test_two_ranks_ddp_with_the_side_stream_on(dev, tmp_path):
    """The configuration the multi-GPU bench runs: S2ANet at a tile size where the packed small levels take the SIDE
    stream (544: P3 = 68 x 68 = 4624 positions > pack_max_positions, P4-P7 packed) while the towers' shared weights
    collect gradients from both streams -- under DDP's bucket hooks.  Replicas bit-identical after every step, the
    update equal to the hand-averaged gradients of a single process, and the autograd engine's stream-mismatch warning
    (an error in the workers) never raised."""
    size = 544
    port = 27000 + os.getpid() % 2000
    out = str(tmp_path / "s.pt")
    mp.spawn(_worker, args=(2, port, "s2anet", False, out, size), nprocs=2, join=True)
    got = torch.load(out)
    ref = _single_process_reference("s2anet", dev, size)
    assert torch.equal(got[0], ref[0])
    for step in range(1, STEPS + 1):
        da, db = got[step] - got[step - 1], ref[step] - ref[step - 1]
        assert float(db.norm()) > 0
        # the yardstick of the eager test (run-to-run spread of the single process itself) at this size: 1e-2 at the
        # first step, then growing as last-bit differences (atomically ordered weight-gradient sums) flip ReLU masks /
        # assignments downstream -- measured 0.5e-2 .. 2.5e-2 at step 3 over four boxes.  A sum instead of a mean, a
        # missing all-reduce or a lost side-stream gradient is off by 50-100 % at step 1.
        assert float((da - db).norm()) <= (1e-2 if step == 1 else 6e-2) * float(db.norm()), (step, float((da - db).norm() / db.norm()))


@pytest.mark.parametrize("name", ["s2anet", "orcnn"])
def test_two_ranks_graph_mode_strict(dev, tmp_path, name):
    """Runner(graph=True) under world_size 2 (round 6): the step is captured per rank, hardened (memset nodes -> kernels)
    and replayed 20 times with one flat all-reduce between the backward graph and the update graph; the replicas must be
    bit-identical after EVERY step (a garbage gradient on one rank -- rounds 4-5 -- breaks that at once), finite, and
    moving."""
    port = 25000 + os.getpid() % 2000 + (7 if name == "orcnn" else 0)
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, port, name, True, out), nprocs=2, join=True)
    got = torch.load(out)
    assert len(got) == REPLAYS + 4 and all(torch.isfinite(h).all() for h in got)
    assert not torch.equal(got[3], got[-1])            # the steps moved the parameters
    norms = [float(h.norm()) for h in got]
    assert max(norms) < 1.5 * norms[0]                 # no blown-up update (the 7e4 parameters of round 5's captured update)


def test_two_ranks_graph_refused_on_request(dev, tmp_path, monkeypatch):
    """JDET_TRAIN_GRAPH_MULTI=0: a graph request under world_size 2 runs eager DDP steps (the behaviour of rounds 4-5)"""
    monkeypatch.setenv("JDET_TRAIN_GRAPH_MULTI", "0")
    port = 27000 + os.getpid() % 2000
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, port, "s2anet", True, out), nprocs=2, join=True)
    got = torch.load(out)
    assert all(torch.isfinite(h).all() for h in got) and not torch.equal(got[3], got[-1])
