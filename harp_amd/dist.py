"""Data-parallel plumbing (new capability: the reference is single-device): frames of a sequence are
sharded over ranks in contiguous blocks, every rank keeps the full parameter arena, and the flat fp32 gradient bucket is
summed with ONE all-reduce per step.  The 1/world factor is applied by the Adam kernel (`grad_scale`), so mean-type losses
average over ranks and frame-independent regularisers (computed identically on every rank from the same RNG seed) are counted
once.

Two transports:
  * `RcclComm` — the production path on a GPU node: RCCL called directly through the C ABI (`harp_allreduce_flat`, csrc/comm.hip)
    on the caller's HIP stream.  It is a plain enqueue, so the collective is a node of the step's hipGraph like every kernel.
    The communicator is bootstrapped with 128 opaque bytes that rank 0 hands to the others over `torch.distributed` (any backend).
  * `torch.distributed.all_reduce` — used when no RcclComm exists (the gloo CPU tests, and N processes sharing one GPU, which RCCL
    refuses: "duplicate GPU")."""
import ctypes
import weakref

import torch
import torch.distributed as dist

from . import _lib


def shard_frames(T, rank, world):
    """contiguous block [lo, hi) of the T frames owned by `rank` (T % world == 0 is required, like C4: 256 = 8 x 32)"""
    if T % world:
        raise ValueError(f"{T} frames do not split evenly over {world} ranks")
    per = T // world
    return rank * per, (rank + 1) * per


def batches(lo, hi, batch_size, step):
    """frame ids of this rank's `step`-th mini-batch: cycles through its block in order"""
    n = hi - lo
    return (torch.arange(batch_size) + step * batch_size) % n + lo


def dist_env():
    """(rank, world) of the default torch.distributed group; (0, 1) when none is initialised"""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def epoch_batches(per, batch, gen):
    """One epoch of a rank's shard: the `per` LOCAL item positions in a fresh random order, cut into batches of `batch` (the last one
    may be shorter) — DataLoader(shuffle=True) (optimize_sequence.py:398) restricted to a shard.  `gen` is a torch.Generator seeded
    identically on every rank, so all ranks draw the SAME within-shard order: every rank runs the same number of steps with the same
    batch sizes (collectives stay matched), and the global batch of step s is the union over ranks r of `lo_r + order[s]` — which a
    single process reproduces exactly with `shards=world` (tests/test_gpu_dist.py)."""
    perm = torch.randperm(int(per), generator=gen)
    return [perm[s0:s0 + batch] for s0 in range(0, int(per), int(batch))]


def mean_over_ranks(value, device=None):
    """Average a host-side scalar (or small tensor) over the ranks of the default group so that every rank continues with the SAME
    number: the epoch loss that drives ReduceLROnPlateau (optimize_sequence.py:581-582) and the finite check must not diverge between
    ranks, or the replicated Adam state does.  All-reduce(sum) hands every rank the identical result; no-op for a single process."""
    rank, world = dist_env()
    if world == 1:
        return value
    t = torch.as_tensor(value, dtype=torch.float64).clone()
    if dist.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t)
    t = (t / world).cpu()
    return t.item() if t.dim() == 0 else t


def ranks_identical(t):
    """True when the tensor is bit-identical on every rank (checksum max == min over the group)"""
    rank, world = dist_env()
    if world == 1:
        return True
    cs = torch.stack((t.double().sum(), t.double().abs().sum(), (t.double() * t.double()).sum())).reshape(3)
    cs = cs if dist.get_backend() == "nccl" else cs.cpu()
    hi, lo = cs.clone(), cs.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return bool((hi == lo).all().item())


def allreduce_flat(bucket, comm=None):
    """sum the flat gradient bucket over all ranks in one collective, in place (no-op for a single process without a communicator)"""
    if comm is not None:
        comm.allreduce(bucket)
    elif dist.is_available() and dist.is_initialized():
        dist.all_reduce(bucket)
    return bucket


def agree(ok, device=None):
    """True only when EVERY rank of the default group passed True (an all-reduce MIN of one int): the way ranks decide together whether to
    take a path that one of them may not be able to take (a rank that went on alone would meet its peers in different collectives)"""
    rank, world = dist_env()
    if world == 1:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend() == "nccl":
        flag = flag.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


def negotiate_comm(device=None, preflight=None, create=None, log=None):
    """The RcclComm of the default process group, or None on EVERY rank when any rank cannot have it.
      1. HARP_RCCL_DEBUG=1: no communicator at all — the caller's steps run eagerly with torch.distributed's all-reduce between the
         backward pass and Adam, outside any hipGraph (the documented way to take RCCL-in-a-graph out of the picture on a new node).
      2. pre-flight, LOCAL (no peer involved, cannot hang): the C-ABI library resolves RCCL (dlopen + dlsym) and ncclGetUniqueId works;
         the ranks agree on the outcome (`agree`) BEFORE anybody enters ncclCommInitRank.
      3. communicator creation (collective: rank 0's 128-byte id over the process group, then ncclCommInitRank); the ranks agree again and
         a rank that did get one destroys it when a peer did not.
    A rank that dies INSIDE ncclCommInitRank or a captured all-reduce still blocks its peers until the process group's timeout
    (init_process_group(timeout=...)) — RCCL's own failure mode.  preflight / create: injectable for the CPU tests."""
    import os
    import sys
    say = log or (lambda m: print(m, file=sys.stderr))
    rank, world = dist_env()
    # the debug switch is folded into the FIRST agreement instead of returning in front of it: a variable set on some ranks only would
    # otherwise leave those ranks out of the collectives their peers enter (every rank takes the same number of them, whatever it holds)
    debug = os.environ.get("HARP_RCCL_DEBUG") == "1"
    ok = not debug
    if debug:
        say(f"[harp_amd.dist] rank {rank}: HARP_RCCL_DEBUG=1 — no RCCL communicator; torch.distributed all-reduce, eager steps")
    else:
        try:
            (preflight or RcclComm.unique_id)()
        except Exception as e:                                       # noqa: BLE001
            say(f"[harp_amd.dist] rank {rank}: RCCL pre-flight failed ({type(e).__name__}: {e})")
            ok = False
    if not agree(ok, device):
        if not debug:
            say(f"[harp_amd.dist] rank {rank}: RCCL pre-flight failed (or HARP_RCCL_DEBUG=1) on some rank — every rank falls back to "
                "torch.distributed's all-reduce")
        return None
    comm = None
    try:
        comm = (create or (lambda: RcclComm.from_process_group(device) if world > 1 else RcclComm.single()))()
    except Exception as e:                                       # noqa: BLE001
        say(f"[harp_amd.dist] rank {rank}: RCCL communicator through the C ABI failed ({type(e).__name__}: {e})")
    if not agree(comm is not None, device):
        if comm is not None:
            comm.destroy()
        say(f"[harp_amd.dist] rank {rank}: no RCCL communicator on some rank — every rank falls back to torch.distributed's all-reduce")
        return None
    return comm


class RcclComm:
    """RCCL communicator owned by the C-ABI library (harp_comm_* / harp_allreduce_flat in include/harp_hip.h)."""

    def __init__(self, rank, world, uid_bytes):
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().harp_comm_create(uid_bytes, int(rank), int(world), ctypes.byref(h)), "harp_comm_create")
        self.handle, self.rank, self.world = h, int(rank), int(world)
        self._engines = weakref.WeakSet()      # FitEngines whose captured step graphs embed this communicator's raw ncclComm_t

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(_lib.lib().harp_comm_unique_id(buf), "harp_comm_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls, device=None):
        """one communicator spanning the default torch.distributed group: rank 0 draws the id, everybody receives it"""
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("RcclComm.from_process_group needs an initialised torch.distributed group (bootstrap channel)")
        rank, world = dist.get_rank(), dist.get_world_size()
        if device is not None:
            torch.cuda.set_device(device)
        # rank 0 broadcasts a failure marker instead of raising in front of the broadcast its peers are waiting in
        box = [None]
        if rank == 0:
            try:
                box = [cls.unique_id()]
            except Exception as e:                                   # noqa: BLE001
                box = [("error", f"{type(e).__name__}: {e}")]
        dist.broadcast_object_list(box, src=0)
        if not isinstance(box[0], (bytes, bytearray)):
            raise RuntimeError(f"RcclComm.from_process_group: rank 0 could not draw the communicator id ({box[0]})")
        return cls(rank, world, box[0])

    @classmethod
    def single(cls):
        """1-rank communicator (tests / HARP_FORCE_DIST): exercises the same RCCL launch path without a bootstrap channel"""
        return cls(0, 1, cls.unique_id())

    def allreduce(self, t, stream=None):
        """in-place sum of a contiguous fp32 HIP tensor, enqueued on `stream` (default: torch's current stream)"""
        if not self.handle:
            raise RuntimeError("RcclComm was destroyed")
        if t.dtype != torch.float32:
            raise TypeError("harp_allreduce_flat reduces float32 buckets")
        _lib.check(_lib.lib().harp_allreduce_flat(self.handle, _lib.ptr(t), t.numel(), _lib.stream() if stream is None else stream),
                   "harp_allreduce_flat")
        return t

    def destroy(self):
        """idempotent; engines that captured this communicator into their step graphs drop those graphs (a replay would hand RCCL a
        dangling ncclComm_t) and fall back to no communicator"""
        if self.handle:
            for eng in list(self._engines):
                if eng.comm is self:
                    eng.set_comm(None)
            h, self.handle = self.handle, ctypes.c_void_p()
            _lib.lib().harp_comm_destroy(h)
