"""Worker of tests/test_gpu_dist.py: one rank of a frame-sharded FitEngine job (launched by torch.distributed.run).

    python -m torch.distributed.run --nproc-per-node 2 ... tests/dist_worker.py OUT.pt STEPS GLOBAL_BATCH

Default: all ranks sit on cuda:0 (HARP_ALL_ON_GPU0-style: the test boxes have ONE GPU, and RCCL refuses two ranks on one device) and
exchange through the gloo process group — i.e. everything of the N > 1 path except the transport: frame sharding, target_offset,
grad_scale = 1/world, same-seed offsets, regularisers counted once, early + final all-reduce, replicated dense Adam.
HARP_WORKER_RCCL=1 (set by test_rccl_ranks_on_real_devices when the box has >= 2 GPUs): one rank per device, the collective is
`harp_allreduce_flat` on an RcclComm built from the process group and captured into every rank's step hipGraph — the production path."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(rank, world, T, S, B, seed=5, device="cuda:0"):
    """the same case on every rank (same seed), then restricted to the rank's frames"""
    from tests._scene import make_fit_case
    case = make_fit_case("hand", T=T, S=S, B=B * world, seed=seed, device=device)     # global-batch engine (B x world frames): renders all targets
    # (also for world == 1: a FRESH engine, so that its texture-offset generator starts at the same counter as the ranks' of an N > 1 job)
    from harp_amd.engine import FitEngine
    from harp_amd import synth
    g = case["eng"]
    tpl = case["tpl"]
    seq = {k: g.params[k].detach().cpu().clone() for k in ("pose", "rot", "trans", "cam")}
    seq["shape"] = g.params["shape"].detach().cpu()[None].repeat(T, 1)
    seq["joints"] = torch.zeros(T, 21, 3)
    eng = FitEngine({k: v.numpy() for k, v in case["model"].items()}, synth.build_topology(tpl["faces0"], 778), tpl["verts_uvs"], tpl["faces_uvs"],
                    case["uv_mask"].float(), seq, S, case["focal"], B, device=device, rank=rank, world_size=world, seed=g.seed)
    with torch.no_grad():
        eng.p_buf.copy_(g.p_buf)
    eng.init_joints = g.init_joints.clone()
    Tl = T // world
    lo = rank * Tl
    eng.set_targets(g.y_true[lo:lo + Tl], g.y_sil[lo:lo + Tl], g.y_sil_col[lo:lo + Tl], frame_offset=lo)
    eng.compute_reference_mesh()
    return case, eng


def main():
    out, steps, global_b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rccl = os.environ.get("HARP_WORKER_RCCL") == "1"
    local = int(os.environ.get("LOCAL_RANK", "0")) if rccl else 0
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    comm = None
    if rccl:
        dist.init_process_group("nccl", device_id=torch.device(device))
        dist.barrier()
    else:
        dist.init_process_group("gloo")
    T, S, B = 4, 128, global_b // world
    case, eng = build(rank, world, T, S, B, device=device)
    eng.keep_image = False
    if rccl:
        from harp_amd.dist import RcclComm
        comm = RcclComm.from_process_group(torch.device(device))
        eng.set_comm(comm)
    Tl = T // world
    lo = rank * Tl
    res = {}
    for it in range(steps):
        fid = (torch.arange(B) + it) % Tl + lo
        if it == 0:
            # gradient arena right after the all-reduce (x grad_scale = the mean over ranks), before Adam touches anything
            eng.fid[:B].copy_(fid.int().to(device)); eng.tfid[:B].copy_((fid - lo).int().to(device))
            eng.set_stage(True, True)
            eng._stage = (True, True)
            eng.forward_backward(True, True, tick=True)
            eng.allreduce()
            torch.cuda.synchronize()
            o, n = eng.opt_span
            res["grad0"] = (eng.g_buf[o:o + n] / world).cpu()
            res["loss0"] = eng.loss_vec.cpu()
            eng.adam(True, True, tick=False)
        else:
            eng.step(fid, True, True)
    torch.cuda.synchronize()
    o, n = eng.opt_span
    res["params"] = eng.p_buf[o:o + n].cpu()
    res["offsets"] = {k: eng.arena.offsets[k][:2] for k in ("pose", "cam", "verts_disps", "shape", "light_positions", "amb_ratio", "texture", "normal_map")}
    res["opt_lo"] = o
    # every rank must hold the same parameters
    cs = eng.p_buf[o:o + n].double().abs().sum().reshape(1)
    cs = cs if rccl else cs.cpu()
    hi, lo_ = cs.clone(), cs.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX); dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
    res["consistent"] = bool((hi == lo_).item())
    res["graph_captured"] = bool(eng._graphs)
    res["transport"] = "rccl (harp_allreduce_flat, graph node)" if comm is not None else "gloo"
    if comm is not None:
        torch.cuda.synchronize()
        comm.destroy()
        assert eng.comm is None and not eng._graphs          # destroying the communicator invalidates the graphs that captured it
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
