"""fused front (csrc/hand_front.hip) and one-launch raster set-up (harp_rasterize_setup2) vs the stand-alone launches"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, bench
kind = sys.argv[1] if len(sys.argv) > 1 else "hand"
eng, focal = bench.build_engine(0, 1, torch.device('cuda'), kind=kind) if kind != "hand" else bench.build_engine(0, 1, torch.device('cuda'))
eng.keep_image = False
B = eng.B
fid = torch.arange(B, dtype=torch.int32, device='cuda')
F, S = eng.topo.F, eng.S
nst = ((S + 63) // 64) ** 2
def split(ws):
    b = ws.view(torch.uint8)
    o = 0
    recs = b[o:o + B * F * 64].view(torch.float32); o += B * F * 64
    bbs = b[o:o + B * F * 16].view(torch.float32); o += B * F * 16
    bins = b[o:o + B * nst * F * 4].view(torch.int32).view(B * nst, F); o += B * nst * F * 4
    c = (B * nst * 4 + 255) // 256 * 256
    cnt = b[o:o + B * nst * 4].view(torch.int32); o += c
    order = b[o:o + B * nst * 4].view(torch.int32); o += c
    nact = b[o:o + 4].view(torch.int32)
    return recs, bbs, bins, cnt, order, nact
def run(front, setup):
    eng.fused_front, eng.fused_setup = front, setup
    eng._lane["fid"][:B].copy_(fid); eng._lane["tfid"][:B].copy_(fid)
    eng.auto_draw = False
    eng.s["ws_c"].zero_(); eng.s["ws_l"].zero_()
    eng.forward_backward(True, True)
    torch.cuda.synchronize()
    s = eng.s
    out = {k: s[k].clone() for k in ("verts_mm", "joints_mm", "vd", "n2", "ndc_c", "ndc_l", "face_c", "face_l", "zl")}
    out["ws_c"] = s["ws_c"].clone(); out["ws_l"] = s["ws_l"].clone()
    out["g"] = eng.g_buf.clone(); out["loss"] = eng._lane["loss_vec"].clone()
    return out
def cmp(a, b, exact_ws):
    for k in a:
        x, y = a[k], b[k]
        if k.startswith("ws_"):
            if not exact_ws: continue
            rx, bx, lx, cx, ox, nx = split(x); ry, by, ly, cy, oy, ny = split(y)
            bad_lists = 0
            for i in range(B * nst):
                n = int(cx[i])
                if n != int(cy[i]) or not torch.equal(lx[i, :n], ly[i, :n]): bad_lists += 1
            same_order_set = torch.equal(torch.sort(ox).values, torch.sort(oy).values)
            # launch order: same multiset, same bucket (count leading zeros) sequence
            bkt = lambda o, c: torch.tensor([32 if c[j] == 0 else 31 - int(c[j]).bit_length() + 1 for j in o.tolist()])
            print("%-6s recs %s bbs %s cnt %s lists_bad %d order perm %s buckets %s nact %d/%d" % (
                k, torch.equal(rx.view(torch.int32), ry.view(torch.int32)), torch.equal(bx.view(torch.int32), by.view(torch.int32)), torch.equal(cx, cy), bad_lists,
                same_order_set, torch.equal(bkt(ox, cx), bkt(oy, cy)), int(nx), int(ny)))
        elif x.dtype == torch.float32:
            d = (x - y).abs().max().item(); r = d / max(x.abs().max().item(), 1e-30)
            print("%-10s max|d| %.3e rel %.3e" % (k, d, r))
        else:
            print("%-10s mismatches %d of %d" % (k, (x != y).sum().item(), x.numel()))
base = run(False, False)
print("== one-launch raster set-up vs three launches (must be identical)"); cmp(base, run(False, True), True)
if eng.fused_front or kind == "hand":
    print("== fused front + set-up vs stand-alone (fp32 rounding)"); cmp(base, run(True, True), False)
