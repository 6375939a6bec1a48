"""world_size-2 gloo test of the data-parallel semantics on CPU: two ranks each compute the step
gradient of THEIR frames (with the CPU oracle standing in for the HIP kernels), sum the flat bucket with
harp_amd.dist.allreduce_flat, scale by 1/world and run Adam — the result must equal one process doing the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from harp_amd.dist import allreduce_flat, batches, shard_frames

KEYS = ("pose", "cam", "verts_disps", "shape", "light_positions", "amb_ratio", "texture", "normal_map")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _grads(sc, fid, P, dists):
    from oracle import harp_ref as H
    with torch.no_grad():
        _, rv = H.prepare_mesh(P, torch.tensor([0]), sc["model"], sc["topo"])
    _, total, _ = H.step_losses(P, fid, sc["model"], sc["topo"], sc["targets"], sc["S"], sc["focal"], rv, dists[0], dists[1])
    for k in KEYS:
        P[k].grad = None
    total.backward()
    return torch.cat([P[k].grad.reshape(-1) for k in KEYS])


def _make(seed=0):
    from tests._scene import make_scene, oracle_params
    sc = make_scene(T=4, S=48, seed=seed)
    src = dict(pose=sc["seq"]["pose"], cam=sc["seq"]["cam"], verts_disps=torch.zeros(3093, 1), shape=sc["seq"]["shape"].mean(0),
               light_positions=torch.tensor(((-0.5, -0.5, -0.5),)).repeat(4, 1), amb_ratio=torch.tensor(0.4),
               texture=torch.full((1, 512, 512, 3), 0.6), normal_map=torch.tensor([0., 0., 1.]).repeat(1, 512, 512, 1),
               rot=sc["seq"]["rot"], trans=sc["seq"]["trans"])
    g = torch.Generator().manual_seed(5)                       # SAME offsets on every rank
    dists = (torch.normal(0, 1.0, (512, 512, 2), generator=g).to(torch.int).long(), torch.normal(0, 2.0, (512, 512, 2), generator=g).to(torch.int).long())
    return sc, oracle_params(sc, src), dists


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, P, dists = _make()
    lo, hi = shard_frames(sc["T"], rank, world)
    fid = batches(lo, hi, 2, 0)
    bucket = _grads(sc, fid, P, dists)
    allreduce_flat(bucket)
    bucket /= world
    if rank == 0:
        out.put(bucket.numpy())
    dist.destroy_process_group()


def test_shard_and_batches():
    assert shard_frames(256, 3, 8) == (96, 128)
    assert batches(96, 128, 32, 0).tolist() == list(range(96, 128))
    assert batches(0, 256, 32, 9).tolist() == list(range(32, 64))
    with pytest.raises(ValueError):
        shard_frames(10, 0, 4)


@pytest.mark.timeout(600)
def test_two_rank_allreduce_equals_full_batch():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = torch.from_numpy(out.get(timeout=500))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sc, P, dists = _make()
    ref = _grads(sc, torch.tensor([0, 1, 2, 3]), P, dists)      # one process, global batch = union of the two shards
    assert (got - ref).norm() <= 2e-4 * ref.norm(), ((got - ref).norm() / ref.norm()).item()
    # frame-independent regularisers are counted once, not `world` times: check the texture block against albedo-only grads


def _sched_worker(rank, world, port, out):
    """two ranks with DIFFERENT local epoch losses: after mean_over_ranks both feed ReduceLROnPlateau the same float, so the learning
    rates stay identical (the replicated-Adam invariant of the data-parallel fit, harp_amd/optimize_sequence.py)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from harp_amd.dist import dist_env, epoch_batches, mean_over_ranks, ranks_identical
    assert dist_env() == (rank, world)
    dummy = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(dummy, patience=1)
    # rank 0 alone would see a plateau from epoch 2 on, rank 1 alone a steady decrease: the mean decides for both
    local = [[1.0, 0.9, 0.9, 0.9, 0.9, 0.9], [1.0, 0.8, 0.6, 0.6, 0.6, 0.6]][rank]
    seen, lrs = [], []
    for v in local:
        m = mean_over_ranks(v)
        seen.append(m)
        sched.step(m)
        lrs.append(dummy.param_groups[0]["lr"])
    gen = torch.Generator().manual_seed(11)
    order = [b.tolist() for b in epoch_batches(6, 4, gen)]
    same = ranks_identical(torch.tensor(lrs)) and ranks_identical(torch.tensor(sum(order, [])).float())
    differ = ranks_identical(torch.tensor([float(rank)]))
    if rank == 0:
        out.put((seen, lrs, order, same, differ))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_epoch_loss_is_averaged_before_the_plateau_scheduler():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sched_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    seen, lrs, order, same, differ = out.get(timeout=200)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert seen == pytest.approx([1.0, 0.85, 0.75, 0.75, 0.75, 0.75])
    assert lrs[:4] == [1e-3] * 4 and lrs[4] == pytest.approx(1e-4)       # two bad epochs (3, 4) > patience 1 -> one decay, on BOTH ranks
    assert same and not differ
    assert sorted(sum(order, [])) == list(range(6)) and [len(b) for b in order] == [4, 2]      # a shuffled shard, ragged last batch


def test_single_process_helpers_are_no_ops():
    from harp_amd.dist import dist_env, mean_over_ranks, ranks_identical
    assert dist_env() == (0, 1) and mean_over_ranks(0.25) == 0.25 and ranks_identical(torch.ones(3))


def test_fit_entry_point_rejects_uneven_shards():
    """optimize_hand_sequence validates the split before anything touches a device: items and the global batch must divide over the shards,
    the shards over the ranks"""
    from harp_amd.optimize_sequence import optimize_hand_sequence
    cfg = {"model_type": "harp"}
    ds = [None] * 8
    for kw in (dict(world_size=3, rank=0), dict(world_size=2, rank=0, shards=3), dict(world_size=1, rank=0, shards=2, batch_size=3)):
        kw.setdefault("batch_size", 4)
        with pytest.raises(ValueError):
            optimize_hand_sequence(cfg, {"pose": torch.zeros(8, 45)}, ds, None, None, None, **kw)
    with pytest.raises(NotImplementedError):
        optimize_hand_sequence({"model_type": "nimble"}, {}, ds, None, None, None)


class _FakeComm:
    def __init__(self):
        self.destroyed = False

    def destroy(self):
        self.destroyed = True


def _negotiate_worker(rank, world, port, out):
    """three rounds of harp_amd.dist.negotiate_comm with injected pre-flight / creation outcomes"""
    from harp_amd.dist import negotiate_comm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res, log = [], []

    def fail():
        raise OSError("librccl.so: cannot open shared object file")
    # 1. the pre-flight fails on rank 1 only: NO rank may enter the (collective) creation
    created = []
    c = negotiate_comm(preflight=(fail if rank == 1 else (lambda: None)), create=lambda: created.append(1) or _FakeComm(), log=log.append)
    res.append((c is None, len(created)))
    # 2. pre-flight fine everywhere, creation fails on rank 0 only: rank 1 gives its communicator back
    mine = []
    def create():
        if rank == 0:
            raise RuntimeError("ncclCommInitRank: unhandled system error")
        mine.append(_FakeComm())
        return mine[0]
    c = negotiate_comm(preflight=lambda: None, create=create, log=log.append)
    res.append((c is None, [m.destroyed for m in mine]))
    # 3. everything fine: every rank keeps its communicator
    c = negotiate_comm(preflight=lambda: None, create=_FakeComm, log=log.append)
    res.append((isinstance(c, _FakeComm) and not c.destroyed,))
    # 4. HARP_RCCL_DEBUG=1: no communicator, nothing attempted
    os.environ["HARP_RCCL_DEBUG"] = "1"
    c = negotiate_comm(preflight=fail, create=fail, log=log.append)
    res.append((c is None,))
    # 5. the variable set on ONE rank only: both ranks still take the same collectives (no hang) and fall back together
    if rank == 1:
        del os.environ["HARP_RCCL_DEBUG"]
    made = []
    c = negotiate_comm(preflight=lambda: None, create=lambda: made.append(1) or _FakeComm(), log=log.append)
    res.append((c is None, len(made)))
    out.put((rank, res, len(log)))
    dist.destroy_process_group()


def test_ranks_fall_back_together_when_one_cannot_have_rccl():
    """harp_amd.dist.negotiate_comm (used by optimize_hand_sequence and bench.py): a rank whose RCCL pre-flight or communicator creation
    fails takes EVERY rank to the torch.distributed fallback — nobody is left alone inside ncclCommInitRank or a captured all-reduce"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_negotiate_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        rank, res, nlog = out.get(timeout=120)
        got[rank] = res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == (True, 0) and got[1][0] == (True, 0)                       # pre-flight failed on rank 1: nobody created anything
    assert got[0][1] == (True, []) and got[1][1] == (True, [True])                  # rank 1's communicator was destroyed again
    assert got[0][2] == (True,) and got[1][2] == (True,)
    assert got[0][3] == (True,) and got[1][3] == (True,)
    assert got[0][4] == (True, 0) and got[1][4] == (True, 0)                       # HARP_RCCL_DEBUG on rank 0 only: nobody created anything
