"""Temporal smoothness terms (mirror of the reference's loss/smooth.py:29-131; SURVEY.md §8f rank 4 — they are not called by
optimize_hand_sequence but are the natural next consumers of the LBS kernel: three hand-layer evaluations per step).

Same classes / methods / arguments.  The three LBS evaluations (previous, current, next frame) run as ONE batched call of the HIP
hand layer (3N frames) instead of three."""
import torch


def _neighbours(fid, n_frames):
    fid_r = torch.where(fid % n_frames == n_frames - 1, fid, fid + 1)          # loss/smooth.py:38-39, 89-90
    fid_l = torch.where(fid % n_frames == 0, fid, fid - 1)
    return fid_l, fid_r


def _joints_lcr(params, fid, mano_layer, use_arm, device, n_frames):
    """(3,N,J,3) joints in mm for (left, current, right) frames through one hand-layer call"""
    N = len(fid)
    fl, fr = _neighbours(fid, n_frames)
    f3 = torch.cat([fl, fid, fr])
    shape = params["shape"].reshape(1, -1).repeat(3 * N, 1).to(device)
    if use_arm:
        _, joints = mano_layer(betas=shape, global_orient=params["rot"][f3].to(device), transl=params["trans"][f3].to(device),
                               right_hand_pose=params["pose"][f3].to(device), return_type="mano_w_arm")
    else:
        _, joints = mano_layer(torch.cat((params["rot"][f3], params["pose"][f3]), 1).to(device), shape, params["trans"][f3].to(device))
    return joints.reshape(3, N, *joints.shape[1:]), (fl, fr)


class LossSmoothPoses:
    def __init__(self, nFrames, use_arm=False):
        self.nFrames, self.norm, self.use_arm = nFrames, "l2", use_arm

    def smooth_pose(self, params, fid, mano_layer, device="cuda"):
        """loss/smooth.py:35-73: root-aligned joints (mm) against the detached mean over the 3-frame window, sum of squares / N"""
        N = len(fid)
        J, _ = _joints_lcr(params, fid, mano_layer, self.use_arm, device, self.nFrames)
        J = J - J[:, :, 0:1]
        interp = ((J[0] + J[1] + J[2]) / 3.0).detach()
        return torch.sum((J[1] - interp) ** 2) / N


class LossSmoothRoots:
    def __init__(self, nFrames, focal_length, res, use_arm=False):
        self.nFrames, self.norm, self.focal_length, self.res, self.use_arm = nFrames, "l2", focal_length, res, use_arm

    def smooth_root(self, params, fid, mano_layer, device="cuda"):
        """loss/smooth.py:86-131: camera translation + detached root joint (m) against the detached 3-frame mean"""
        N = len(fid)
        J, (fl, fr) = _joints_lcr(params, fid, mano_layer, self.use_arm, device, self.nFrames)
        roots = []
        for j, f in zip(J, (fl, fid, fr)):
            cam = params["cam"][f].to(device)
            t = torch.stack([cam[:, 1], cam[:, 2], 2 * self.focal_length / (self.res * cam[:, 0] + 1e-9)], dim=1)
            roots.append(t + j[:, 0].detach() / 1000.0)
        interp = ((roots[0] + roots[1] + roots[2]) / 3.0).detach()
        return torch.sum((roots[1] - interp) ** 2) / N
