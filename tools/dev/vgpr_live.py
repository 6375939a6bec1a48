"""Approximate VGPR liveness profile of one kernel in an AMDGPU .s file (linear backward scan, branches ignored): prints the number of
live VGPRs every N instructions together with the nearest preceding source-line marker, to see WHERE a kernel's register peak is.
    python tools/dev/vgpr_live.py file.s kernel_name_substring [step]"""
import re
import sys

txt = open(sys.argv[1]).read()
name = sys.argv[2]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 100
m = re.search(r"^(\S*%s\S*):" % re.escape(name), txt, re.M)
start = m.end()
end = txt.index('.Lfunc_end', start)
lines = [l.strip() for l in txt[start:end].split('\n')]
ins = [(i, l) for i, l in enumerate(lines) if l and not l.startswith(('.', ';', '//')) and not l.endswith(':')]


def regs(tok):
    out = []
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', tok):
        out += list(range(int(a), int(b) + 1))
    out += [int(x) for x in re.findall(r'\bv(\d+)\b', tok)]
    return out


live = set()
prof = []
for idx in range(len(ins) - 1, -1, -1):
    _, l = ins[idx]
    parts = l.split(None, 1)
    op = parts[0]
    ops = parts[1].split(',') if len(parts) > 1 else []
    ops = [o.strip() for o in ops]
    if not ops:
        prof.append(len(live)); continue
    stores = op.startswith(('global_store', 'global_atomic', 'ds_write', 'ds_add', 'ds_cmpst', 'scratch_store', 'buffer_store', 'ds_min', 'ds_max', 's_', 'v_cmp', 'v_cmpx'))
    returning = ('_rtn' in op) or op.startswith(('ds_read', 'ds_bpermute', 'ds_cmpst_rtn'))
    if stores and not returning:
        dst, src = [], ops
    else:
        dst, src = [ops[0]], ops[1:]
        if op in ('v_fmac_f32_e32', 'v_fmac_f32_e64', 'v_pk_fmac_f16', 'v_writelane_b32', 'v_mac_f32_e32') or 'dpp' in l or 'sdwa' in l:
            src = ops          # destination is also read
    for d in dst:
        for r in regs(d):
            live.discard(r)
    for s_ in src:
        for r in regs(s_):
            live.add(r)
    prof.append(len(live))
prof = prof[::-1]
print('instructions', len(ins), 'peak live', max(prof), 'at', prof.index(max(prof)))
for k in range(0, len(ins), step):
    seg = prof[k:k + step]
    print(f'{k:5d}  max {max(seg):3d}  ' + ins[k + seg.index(max(seg))][1][:90])

# ---- what is live at the peak: the defining instruction of every live register (nearest definition above the peak)
if len(sys.argv) > 4 and sys.argv[4] == "peak":
    pk = prof.index(max(prof))
    live = set()
    for idx in range(len(ins) - 1, pk, -1):          # recompute the live set just below the peak
        _, l = ins[idx]
        parts = l.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
        if not ops:
            continue
        stores = op.startswith(('global_store', 'global_atomic', 'ds_write', 'ds_add', 'ds_cmpst', 'scratch_store', 'buffer_store', 'ds_min', 'ds_max', 's_', 'v_cmp', 'v_cmpx'))
        returning = ('_rtn' in op) or op.startswith(('ds_read', 'ds_bpermute', 'ds_cmpst_rtn'))
        if stores and not returning:
            dst, src = [], ops
        else:
            dst, src = [ops[0]], ops[1:]
            if op in ('v_fmac_f32_e32', 'v_fmac_f32_e64', 'v_writelane_b32') or 'dpp' in l:
                src = ops
        for d in dst:
            for r in regs(d):
                live.discard(r)
        for s_ in src:
            for r in regs(s_):
                live.add(r)
    import collections
    by = collections.defaultdict(list)
    for r in sorted(live):
        d = None
        for idx in range(pk, -1, -1):
            parts = ins[idx][1].split(None, 1)
            if len(parts) > 1 and r in regs(parts[1].split(',')[0]) and not parts[0].startswith(('global_store', 'global_atomic', 'ds_write', 'ds_add', 's_', 'v_cmp')):
                d = (idx, parts[0])
                break
        by[d].append(r)
    for d, rs in sorted(by.items(), key=lambda kv: (kv[0] or (0, ''))):
        print(d, rs, ins[d[0]][1][:100] if d else '')
