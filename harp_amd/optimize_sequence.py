"""Mirror of the fitting API of the reference's optimize_sequence.py: get_mesh_subdivider (:67-89), init_params (:181-250),
get_optimizers (:253-310) and optimize_hand_sequence (:313-816, loop body :446-582).

`optimize_hand_sequence` keeps the reference's signature and schedule (stages, batch 18, shuffle, dense Adam groups and
learning rates, ReduceLROnPlateau(patience=40) on the coarse group, checkpoint format) but runs every step through the fused
engine (harp_amd/engine.py) on HBM-resident targets.  The per-op API (`prepare_mesh`, `render_image`, losses) stays available
for callers that drive autograd themselves (`visualize_val`-style code)."""
import numpy as np
import torch

from .engine import FitEngine, LOSS_NAMES
from .synth import build_topology
from .utils import file_utils
from .utils.data_util import ResidentTargets
from .utils.visualize import MeshSubdivider


def get_mesh_subdivider(hand_layer, use_arm=False, device="cuda"):
    """optimize_sequence.py:67-89"""
    if use_arm:
        return MeshSubdivider(hand_layer.right_arm_faces_tensor, 1026, device)
    return MeshSubdivider(hand_layer.th_faces, 778, device)


def load_uv_mask(configs, uv_size):
    """optimize_sequence.py:174-178"""
    from PIL import Image
    uv_mask_pil = Image.open(configs["uv_mask"]).convert("L").resize(uv_size)
    return torch.tensor(np.asarray(uv_mask_pil) / 255)


def init_params(input_params, VERT_DISPS, VERT_DISPS_NORMALS, VERTS_COLOR, mano_faces, verts_textures, VERTS_UVS=None, FACES_UVS=None,
                model_type="harp", use_arm=False, configs=None, device="cuda", uv_mask=None):
    """optimize_sequence.py:181-250: the same dict (keys, shapes, initial values).  All leaves live on `device` (the reference keeps
    most of them on the CPU and copies them every iteration)."""
    if model_type != "harp" or verts_textures:
        raise NotImplementedError("model_type 'harp' with UV textures is the path in scope (SURVEY.md §8)")
    P = lambda t: torch.nn.Parameter(t.detach().clone().float().to(device), requires_grad=True)
    params = {}
    params["trans"], params["pose"], params["rot"] = P(input_params["trans"]), P(input_params["pose"]), P(input_params["rot"])
    params["shape"] = P(input_params["shape"].mean(dim=0))
    params["wrist_pose"] = P(torch.zeros([params["pose"].shape[0], 3]))
    params["init_joints"] = input_params["joints"]
    n_mesh_verts = 4083 if use_arm else 3093
    params["verts_disps"] = P(torch.zeros(n_mesh_verts, 1 if (VERT_DISPS_NORMALS or not VERT_DISPS) else 3))
    verts_rgb_init = torch.from_numpy(VERTS_COLOR) if VERTS_COLOR is not None else torch.ones(778, 3)
    params["verts_rgb"] = P(verts_rgb_init)
    params["verts_uvs"], params["faces_uvs"] = VERTS_UVS, FACES_UVS
    params["texture"] = P(torch.tensor([232, 190, 172]).repeat(1, 512, 512, 1) / 255.)
    params["uv_mask"] = uv_mask if uv_mask is not None else load_uv_mask(configs, params["texture"].shape[1:3])
    params["normal_map"] = P(torch.tensor([0.0, 0.0, 1.0]).repeat(1, 512, 512, 1))
    total_frame = input_params["cam"].shape[0]
    params["light_positions"] = P(torch.tensor(((-0.5, -0.5, -0.5),)).repeat(total_frame, 1))
    params["amb_ratio"] = P(torch.tensor(0.4))
    params["mesh_faces"] = mano_faces
    params["cam"] = P(input_params["cam"])
    return params


def get_optimizers(params, configs):
    """optimize_sequence.py:253-310 (torch optimisers over the parameter dict, for callers using the autograd API)."""
    pose_params = [params["pose"], params["cam"]]
    shape_params = [params["verts_disps"], params["shape"]] if configs["use_vert_disp"] else [params["shape"]]
    groups = [{"params": pose_params, "lr": 1.0e-3}]
    if configs["use_arm"] and configs["opt_arm_pose"]:
        groups.append({"params": [params["wrist_pose"], params["rot"]], "lr": 1.0e-3})
    if not configs["known_appearance"]:
        groups.append({"params": shape_params, "lr": 1.0e-3})
    opt_coarse = torch.optim.Adam(groups)
    app = [params["light_positions"], params["amb_ratio"]]
    if not configs["known_appearance"]:
        app += [params["texture"], params["normal_map"]]
    opt_app = torch.optim.Adam(app, lr=1.0e-2)
    sched_coarse = torch.optim.lr_scheduler.ReduceLROnPlateau(opt_coarse, patience=40)
    return opt_coarse, opt_app, sched_coarse


def stage_flags(epoch_id, training_stage):
    """optimize_sequence.py:507-515 -> (COARSE_OPT, APP_OPT)"""
    if epoch_id < training_stage[0]:
        return True, False
    if epoch_id < training_stage[0] + training_stage[1]:
        return True, True
    return False, True


def optimize_hand_sequence(configs, input_params, images_dataset, val_params, val_images_dataset, hand_layer,
                           VERTS_UVS=None, FACES_UVS=None, VERTS_COLOR=None, device="cuda", uv_mask=None, batch_size=18, log_fn=None,
                           seed=0, vgg=None, rank=None, world_size=None, shards=None, plateau_patience=40, plateau_threshold=1e-4, device_schedule=True):
    """Fit the sequence (optimize_sequence.py:313-596).  Returns the parameter dict in the reference's checkpoint layout.
    `images_dataset[i]` -> (fid, y_true (S,S,3), y_sil (S,S,1), y_sil_eroded (S,S,1)) like utils/data_util.ImagesDataset.

    Data-parallel (the reference is single-device): launched under `torch.distributed.run` (or with rank / world_size given)
    every rank calls this function with the SAME arguments.  The dataset's items are cut into `shards` (default: world) contiguous
    shards, a rank decodes and keeps resident only its own (`ResidentTargets(frames=...)`), `batch_size` stays the GLOBAL batch of the
    reference (each shard contributes batch_size / shards frames per step), the flat gradient bucket is summed over ranks before the
    replicated dense Adam step (RCCL through `harp_allreduce_flat` inside the step's hipGraph when the process group is nccl = one
    device per rank; torch.distributed otherwise), the per-epoch shuffle comes from the shared seed WITHIN each shard
    (harp_amd.dist.epoch_batches), the epoch loss is averaged over ranks BEFORE ReduceLROnPlateau sees it (:581-582) so that every rank
    takes the same learning-rate decision, the finite check runs on that averaged loss, and only rank 0 writes checkpoints.
    `shards > world` makes one process walk several shards per step — a 1-rank job with shards = N visits the same global batches as an
    N-rank job.  plateau_patience / plateau_threshold: ReduceLROnPlateau's arguments (reference: patience 40, default threshold)."""
    from . import dist as hdist
    if configs["model_type"] != "harp":
        raise NotImplementedError("only model_type 'harp' (SURVEY.md §8: 'html' / 'nimble' are out of scope)")
    env_rank, env_world = hdist.dist_env()
    rank = env_rank if rank is None else int(rank)
    world = env_world if world_size is None else int(world_size)
    shards = world if shards is None else int(shards)
    n_items = len(images_dataset)
    if shards % world or n_items % shards or batch_size % shards:
        raise ValueError(f"{n_items} dataset items / global batch {batch_size} do not split evenly over {shards} shards on {world} ranks")
    k, per, b = shards // world, n_items // shards, batch_size // shards          # shards per rank, items per shard, frames per shard and step
    lo = rank * k * per                                                           # this rank's items: [lo, lo + k * per)
    S, T = configs["img_size"], input_params["pose"].shape[0]
    use_arm = bool(configs["use_arm"])
    faces0 = np.asarray((hand_layer.right_arm_faces_tensor if use_arm else hand_layer.th_faces).detach().cpu())
    topo = build_topology(faces0, 1026 if use_arm else 778)
    if uv_mask is None:
        uv_mask = load_uv_mask(configs, (512, 512))
    eng = FitEngine(hand_layer._model_np, topo, torch.as_tensor(VERTS_UVS).reshape(-1, 2), torch.as_tensor(FACES_UVS).reshape(-1, 3),
                    torch.as_tensor(uv_mask).float(), input_params, S, configs["focal_length"], b * k, device=device,
                    self_shadow=configs["self_shadow"], share_light_position=configs["share_light_position"], seed=seed,
                    use_arm=use_arm, opt_arm_pose=bool(configs.get("opt_arm_pose", False)), rank=rank, world_size=world)
    rt = ResidentTargets(images_dataset, frames=range(lo, lo + k * per))         # decoded once, resident in HBM (utils/data_util.py)
    if int(rt.fid.min()) < 0 or int(rt.fid.max()) >= T:
        raise ValueError(f"dataset frame ids span [{int(rt.fid.min())}, {int(rt.fid.max())}] but the parameter tables hold {T} frames")
    eng.set_targets(*rt.tensors())
    comm = None
    if world > 1:
        import torch.distributed as tdist
        if not (tdist.is_available() and tdist.is_initialized()):
            raise RuntimeError("world_size > 1 needs an initialised torch.distributed process group (torch.distributed.run)")
        if tdist.get_backend() == "nccl":                                # one device per rank: RCCL straight from the step's hipGraph
            # pre-flight + agreement over the process group: either every rank gets the communicator or none does (then the steps run
            # eagerly with torch.distributed's all-reduce); HARP_RCCL_DEBUG=1 forces that fallback (harp_amd.dist.negotiate_comm)
            comm = hdist.negotiate_comm(torch.device(device))
            if comm is not None:
                eng.set_comm(comm)
    # perceptual term (:404-405, :546-547): needs the pretrained VGG16 filters, which cannot be downloaded here — pass a ready module
    # (`vgg=`) or the path of torchvision's vgg16 state dict (configs["vgg_weights"]); without either the term is left out
    if vgg is None and configs.get("vgg_weights"):
        from .model.vgg import Vgg16Features
        vgg = Vgg16Features(layers_weights=[1, 1 / 16, 1 / 8, 1 / 4, 1], weights=configs["vgg_weights"])
    if vgg is not None:
        # configs["vgg_precision"]: 0 (default) float32 MFMA — a float32 fma chain like the reference's fp32 convolutions; 1 three-term bf16
        # split (2.5x faster, ~16 mantissa bits per product; the reference's own stack allows TF32 here).  configs["vgg_cache_bytes"]: HBM the
        # cached target activations may take (default 128 GB: 307 MB per 512x512 frame for the bounded mode, FitEngine.set_perceptual)
        eng.set_perceptual(vgg, weight=1.0, precision=int(configs.get("vgg_precision", 0)), cache_bytes=int(configs.get("vgg_cache_bytes", 128 << 30)))
    eng.keep_image = False                                           # the fused L1 consumes y_pred in the shader; nothing reads the image back
    eng.accumulate_loss = True
    eng.lean_app_stage = True                                        # the appearance-only stage steps opt_app alone (:264-310, :567-573): no geometry gradients are formed for it
    if configs["start_from"]:
        restore_checkpoint(eng, configs, input_params)
    if configs["known_appearance"]:
        # optimize_sequence.py:264-289: shape / displacement leave opt_coarse, texture / normal map leave opt_app
        eng.frozen = ("verts_disps", "shape", "texture", "normal_map")
        # ... and the key-point anchor and the mesh regularisers are not part of a test sequence's objective (:523, :531)
        eng.set_disabled_terms(("kps_anchor", "vert_disp_reg", "laplacian", "normal", "arap"))
    # ReduceLROnPlateau lives on the host; torch's own scheduler drives a dummy optimiser and the lr is mirrored to the device
    dummy = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(dummy, patience=plateau_patience, threshold=plateau_threshold)
    gen = torch.Generator().manual_seed(seed)                        # the SAME stream of draws on every rank
    # (the loss weights of :411-422 are the engine's LOSS_WEIGHTS: its kernels scale the gradients with them and its step epilogue forms sum_loss)
    own = torch.arange(k) * per                                      # first local row of each of this rank's shards
    try:
        def draw_epoch():
            """the next epoch's batches (DataLoader(shuffle=True) over the DATASET's items, :398) and — its full batches as ONE device schedule
            (parameter rows = the items' own fids, :446, :464; target rows = the items): every such step is a bare graph replay that fetches
            its own row, no host tensor per step like the DataLoader's (:399); same shape every epoch, so the captured step graphs keep
            reading the same two buffers — the schedule to hand to eng.set_schedule"""
            batches = hdist.epoch_batches(per, b, gen)
            items = [(own[:, None] + order[None, :]).reshape(-1) for order in batches]  # local rows of the resident targets, shard by shard
            full = [it for it in items if it.numel() == eng.B] if device_schedule else []
            rows = torch.stack(full) if full else None
            return items, (None if rows is None else (rt.fid[rows], rows))
        items, sch = draw_epoch()
        if sch is not None:
            eng.set_schedule(sch[0], tschedule=sch[1])
        eng.loss_total.zero_()            # the engine adds every step's sum_loss (:553-559) to it on the device: no per-step host arithmetic, no sync
        for epoch_id in range(configs["total_epoch"]):
            coarse, app = stage_flags(epoch_id, configs["training_stage"])
            for item in items:
                if item.numel() == eng.B and device_schedule:
                    eng.step(None, coarse, app)
                else:
                    eng.step(rt.fid[item], coarse, app, tfid=item)                     # the last, partial batch (:396-399): explicit rows, a graph of its own size
            nb = len(items)
            # one sync per epoch; N > 1: the mean over ranks (image terms are means over a rank's frames, regularisers are identical),
            # the same float on every rank
            # ... and everything of the NEXT epoch that does not depend on this epoch's loss — its shuffle, its schedule (stream-ordered behind
            # the steps above) — is enqueued before that sync, so the device goes from this epoch's last step to the next one's first
            total = eng.loss_total.clone()
            eng.loss_total.zero_()
            if epoch_id + 1 < configs["total_epoch"]:
                items, sch = draw_epoch()
                if sch is not None:
                    eng.set_schedule(sch[0], tschedule=sch[1])
            epoch_loss = float(total.item()) / nb
            mean_loss = float(hdist.mean_over_ranks(epoch_loss, device=eng.dev)) if world > 1 else epoch_loss
            if not np.isfinite(mean_loss):
                raise FloatingPointError(f"non-finite loss at epoch {epoch_id}")      # the reference drops into pdb (:525-527); all ranks raise together
            if coarse:
                sched.step(mean_loss)                                                  # :581-582
                eng.set_lr(lr_coarse=dummy.param_groups[0]["lr"])
            if log_fn is not None:
                log_fn(epoch_id, mean_loss, eng)
            if epoch_id % 200 == 0 and epoch_id > 0 and rank == 0:
                file_utils.save_result(export_params(eng, input_params, VERTS_UVS, FACES_UVS, uv_mask, hand_layer), configs["base_output_dir"],
                                       test=configs["known_appearance"])               # :590-591
        params = export_params(eng, input_params, VERTS_UVS, FACES_UVS, uv_mask, hand_layer)
        if rank == 0:
            file_utils.save_result(params, configs["base_output_dir"], test=configs["known_appearance"])     # :595-596
    finally:
        if comm is not None:
            torch.cuda.synchronize(eng.dev)
            comm.destroy()                                           # drops the step graphs that captured it
    return params


def main(argv=None):
    """optimize_sequence.py:819-838 over this package's loaders — and the data-parallel launch:

        python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m harp_amd.optimize_sequence --config cfg.yaml

    one rank per GPU (LOCAL_RANK), nccl (= RCCL) process group; HARP_ALL_ON_GPU0=1 puts every rank on cuda:0 over gloo (one-GPU boxes)."""
    import argparse
    import os
    import yaml
    import torch.distributed as tdist
    from .utils import hand_model_utils
    from .utils.config_utils import get_config
    from .utils.data_util import load_multiple_sequences
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, help="yaml with the keys of utils/config_utils.get_config")
    ap.add_argument("--batch-size", type=int, default=18)
    args = ap.parse_args(argv)
    with open(args.config) as f:
        configs = get_config(write_yaml=False, **yaml.safe_load(f))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    shared = os.environ.get("HARP_ALL_ON_GPU0") == "1"
    device = "cuda:0" if shared else f"cuda:{local}"
    torch.cuda.set_device(device)
    if world > 1:
        import datetime
        timeout = datetime.timedelta(seconds=int(os.environ.get("HARP_DIST_TIMEOUT_S", "600")))     # a dead peer ends the job instead of hanging it
        if shared:
            tdist.init_process_group("gloo", timeout=timeout)
        else:
            tdist.init_process_group("nccl", device_id=torch.device(device), timeout=timeout)
    configs["device"] = device
    hand_layer, VERTS_UVS, FACES_UVS, VERTS_COLOR = hand_model_utils.load_hand_model(configs)
    mano_params, images_dataset, val_mano_params, val_images_dataset = load_multiple_sequences(
        configs["metro_output_dir"], configs["image_dir"], train_list=configs["train_list"], val_list=configs["val_list"],
        average_cam_sequence=configs["average_cam_sequence"], use_smooth_seq=configs["use_smooth_seq"], model_type=configs["model_type"])
    params = optimize_hand_sequence(configs, mano_params, images_dataset, val_mano_params, val_images_dataset, hand_layer, VERTS_UVS, FACES_UVS,
                                    VERTS_COLOR, device=device, batch_size=args.batch_size)
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()
    return params


def restore_checkpoint(eng, configs, input_params):
    """Resume from `configs["start_from"]/saved_params[_test].pkl` the way optimize_sequence.py:355-389 does — including its
    re-initialisation: the pose track is re-interpolated linearly between every 30th frame, `trans` and `rot` are replaced by their
    sequence means, parameters missing from older checkpoints get the init_params defaults; with known_appearance (and not
    pose_already_opt) pose / trans / rot / cam restart from the METRO input instead."""
    known, posed = bool(configs["known_appearance"]), bool(configs["pose_already_opt"])
    ck = file_utils.load_result(configs["start_from"], device="cpu", test=known and posed)
    ck = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in ck.items()}
    if known and not posed:
        for k in ("trans", "pose", "rot", "cam"):
            ck[k] = input_params[k].detach().clone()
    T = eng.params["pose"].shape[0]
    if ck["pose"].shape[0] != T:
        raise ValueError(f"checkpoint holds {ck['pose'].shape[0]} frames, the sequence has {T}")
    pose = ck["pose"].clone()
    for i in range(T // 30 - 1):
        for j in range(30):
            pose[i * 30 + j] = ((30 - j) * ck["pose"][i * 30] + j * ck["pose"][i * 30 + 30]) / 30.0
    ck["pose"] = pose
    ck["trans"] = torch.zeros_like(ck["trans"]) + ck["trans"].mean(0)
    ck["rot"] = torch.zeros_like(ck["rot"]) + ck["rot"].mean(0)
    with torch.no_grad():
        for k in ("trans", "pose", "rot", "shape", "wrist_pose", "verts_disps", "texture", "normal_map", "light_positions", "amb_ratio", "cam"):
            if k in ck and ck[k] is not None:                     # absent keys keep the init_params defaults the engine starts with
                eng.params[k].copy_(torch.as_tensor(ck[k]).to(eng.dev).reshape(eng.params[k].shape))
    eng.compute_reference_mesh()                                  # ARAP reference = frame 0 under the restored parameters (:429-435)


def export_params(eng, input_params, VERTS_UVS, FACES_UVS, uv_mask, hand_layer):
    """the reference's parameter dict (init_params keys) from the engine's arena"""
    out = {k: eng.params[k].detach().clone() for k in ("trans", "pose", "rot", "shape", "wrist_pose", "verts_disps", "texture", "normal_map",
                                                       "light_positions", "amb_ratio", "cam")}
    out.update(init_joints=input_params["joints"], verts_rgb=torch.ones(778, 3), verts_uvs=VERTS_UVS, faces_uvs=FACES_UVS,
               uv_mask=torch.as_tensor(uv_mask), mesh_faces=getattr(hand_layer, "right_arm_faces_tensor", getattr(hand_layer, "th_faces", None)))
    return out


if __name__ == "__main__":
    main()
