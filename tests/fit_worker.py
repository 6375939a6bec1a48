"""Worker of tests/test_gpu_dist.py::test_data_parallel_fit_*: one rank of `optimize_hand_sequence` launched by torch.distributed.run.

    python -m torch.distributed.run --nproc-per-node 2 ... tests/fit_worker.py OUT.pt SHARDS

All ranks sit on cuda:0 and talk through gloo by default (the test boxes have ONE GPU; RCCL refuses two ranks on one device);
HARP_WORKER_RCCL=1: one rank per device over nccl, i.e. the RcclComm path of the fitting API.  Every rank builds the same seeded
scene and calls the fitting entry point with the same arguments — sharding, the shared-seed shuffle, the all-reduced epoch loss in
front of ReduceLROnPlateau and the rank-0 checkpoint are the function's own business (harp_amd/optimize_sequence.py)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, shards = sys.argv[1], int(sys.argv[2])
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    rccl = os.environ.get("HARP_WORKER_RCCL") == "1"
    local = int(os.environ.get("LOCAL_RANK", "0")) if rccl else 0
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    if world > 1:
        if rccl:
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group("gloo")
    from harp_amd.dist import ranks_identical
    from harp_amd.manopth.manolayer import ManoLayer
    from harp_amd.optimize_sequence import optimize_hand_sequence
    from harp_amd.utils.config_utils import get_config
    from tests._scene import make_scene
    T, S = 8, 96
    sc = make_scene(T=T, S=S, seed=3)
    outdir = out + f".dir{rank}/"                      # only rank 0 may write into its directory
    os.makedirs(outdir, exist_ok=True)
    cfg = get_config(write_yaml=False, use_arm=False, img_size=S, focal_length=sc["focal"], total_epoch=4, training_stage=[2, 1, 1],
                     base_output_dir=outdir)
    layer = ManoLayer(flat_hand_mean=False, use_pca=False, model=sc["model_np"], device=device)
    tg = sc["targets"]
    ds = [(i, tg["y_true"][i], tg["y_sil"][i][..., None], tg["y_sil_col"][i][..., None]) for i in range(T)]
    hist, engs = [], []

    def log(epoch, loss, eng):
        h = eng.hyper.cpu().numpy().view(eng.hyper_np.dtype)
        hist.append((epoch, loss, float(h["lr"][0]), float(h["lr"][1])))
        engs[:] = [eng]
    # global batch 4 = 2 frames of each of the 2 shards per step, 2 steps per epoch; patience 0 + a threshold no epoch can meet: the
    # coarse learning rate decays after the second epoch — driven by the epoch loss every rank must agree on
    params = optimize_hand_sequence(cfg, sc["seq"], ds, None, None, layer, torch.from_numpy(sc["tpl"]["verts_uvs"])[None],
                                    torch.from_numpy(sc["tpl"]["faces_uvs"])[None], device=device, uv_mask=sc["uv_mask"], batch_size=batch,
                                    log_fn=log, shards=shards, plateau_patience=0, plateau_threshold=0.5)
    eng = engs[0]
    o, n = eng.opt_span
    res = dict(hist=hist, params=eng.p_buf[o:o + n].cpu(), opt_lo=o, world=world, batch=eng.B, rows=int(eng.y_true.shape[0]),
               offsets={k: eng.arena.offsets[k][:2] for k in ("pose", "cam", "verts_disps", "shape", "light_positions", "amb_ratio", "texture", "normal_map")},
               identical=ranks_identical(eng.p_buf) and ranks_identical(eng.m_buf) and ranks_identical(eng.v_buf),
               wrote=os.path.exists(outdir + "saved_params.pkl"), graphs=len(eng._graphs), comm=type(eng.comm).__name__,
               transport=("rccl" if rccl else "gloo") if world > 1 else "none")
    if world > 1:
        flags = [None] * world
        dist.all_gather_object(flags, res["wrote"])
        res["wrote_by_rank"] = flags
    if rank == 0:
        torch.save(res, out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
