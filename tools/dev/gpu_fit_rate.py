"""steps / s of the fitting API itself (harp_amd.optimize_sequence.optimize_hand_sequence) at the bench's size: 64 frames at 512^2, batch 32,
all three stages; per-epoch wall time from the log callback (one device sync per epoch).  python tools/dev/gpu_fit_rate.py"""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import tempfile
import torch
from harp_amd.manopth.manolayer import ManoLayer
from harp_amd.optimize_sequence import optimize_hand_sequence
from harp_amd.utils.config_utils import get_config
from tests._scene import make_scene
T, S, B = 256, 512, 32
dev_sched = os.environ.get('HARP_FIT_EXPLICIT') != '1'
sc = make_scene(T=T, S=S, seed=3)
out = tempfile.mkdtemp() + "/"
E = 6
cfg = get_config(write_yaml=False, use_arm=False, img_size=S, focal_length=sc["focal"], total_epoch=3 * E, training_stage=[E, E, E], base_output_dir=out)
layer = ManoLayer(flat_hand_mean=False, use_pca=False, model=sc["model_np"], device="cuda")
tg = sc["targets"]
ds = [(i, tg["y_true"][i], tg["y_sil"][i][..., None], tg["y_sil_col"][i][..., None]) for i in range(T)]
stamps = []
engs = []
def log(epoch, loss, eng):
    torch.cuda.synchronize(); stamps.append((epoch, time.perf_counter(), loss)); engs[:] = [eng]
optimize_hand_sequence(cfg, sc["seq"], ds, None, None, layer, torch.from_numpy(sc["tpl"]["verts_uvs"])[None], torch.from_numpy(sc["tpl"]["faces_uvs"])[None],
                       device="cuda", uv_mask=sc["uv_mask"], batch_size=B, log_fn=log, device_schedule=dev_sched)
steps = T // B
for s0 in range(3):
    ep = [stamps[i][1] - stamps[i - 1][1] for i in range(s0 * E + 2, (s0 + 1) * E)]       # (the first epochs of a stage capture its graph)
    print(("device schedule " if dev_sched else "explicit batches ") + "stage %d: %.3f ms / step (median epoch of %d steps, %d frames / step), loss %.5f" % (s0, sorted(ep)[len(ep) // 2] / steps * 1e3, steps, B, stamps[(s0 + 1) * E - 1][2]))
# the same engine, same scene, bare scheduled steps (what the epoch loop could reach without any per-epoch work)
eng = engs[0]
for name, (c, a) in (("stage 0", (True, False)), ("stage 1", (True, True)), ("stage 2", (False, True))):
    for _ in range(10): eng.step(None, c, a)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100): eng.step(None, c, a)
    torch.cuda.synchronize(); print("bare engine steps, %s: %.3f ms / step" % (name, (time.perf_counter() - t) / 100 * 1e3))
