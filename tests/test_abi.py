"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol include/harp_hip.h
declares (no compute calls without a GPU); the ctypes mirrors match the C structs; the product has no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from harp_amd import build, _lib
    build.build(force=False, verbose=False)
    return _lib.lib()


def test_header_symbols_exported(lib):
    from harp_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "harp_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t)\s+(harp_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in harp_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_struct_layouts_match_c(tmp_path):
    """compile a tiny C program against the header and compare sizeof/offsetof with the ctypes mirrors"""
    import subprocess
    from harp_amd import _lib
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "harp_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(harp_shade_args), offsetof(harp_shade_args, B), offsetof(harp_shade_args, rgb), sizeof(harp_mano_model),'
                   'sizeof(harp_frame_tables), sizeof(harp_adam_hyper), offsetof(harp_frame_tables, share_light), sizeof(harp_mesh_chain),'
                   'sizeof(harp_hand_front), offsetof(harp_hand_front, step), sizeof(harp_step_frame), offsetof(harp_step_frame, draw_counter),'
                   'sizeof(harp_tree_model), sizeof(harp_arm_front), offsetof(harp_arm_front, weights_T), offsetof(harp_arm_front, step));'
                   'return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(_lib.ShadeArgs), _lib.ShadeArgs.B.offset, _lib.ShadeArgs.rgb.offset, ctypes.sizeof(_lib.ManoModel),
            ctypes.sizeof(_lib.FrameTables), 32, _lib.FrameTables.share_light.offset, ctypes.sizeof(_lib.MeshChain),
            ctypes.sizeof(_lib.HandFront), _lib.HandFront.step.offset, ctypes.sizeof(_lib.StepFrame), _lib.StepFrame.draw_counter.offset,
            ctypes.sizeof(_lib.TreeModel), ctypes.sizeof(_lib.ArmFront), _lib.ArmFront.weights_T.offset, _lib.ArmFront.step.offset]
    assert got == want, (got, want)


def test_no_cpu_fallback(lib):
    from harp_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.rasterize_fwd(torch.zeros(1, 3, 3), torch.zeros(1, 3, dtype=torch.int32), 8)
    # records | bboxes | lists | counts, order, nact (256 B each) | hit bitmaps (one 64-bit word per frame, super-tile and 64 faces)
    assert lib.harp_rasterize_ws_bytes(2, 10, 64) == 2 * 10 * 64 + 2 * 10 * 16 + 2 * 1 * 10 * 4 + 3 * 256 + 2 * 1 * 1 * 8      # pure host arithmetic
    assert lib.harp_rasterize_fwd(None, None, 1, 1, 1, 8, 0, 0.0, 1.0, None, None, None, None, None) == 1   # HARP_ERR_ARG, no launch


def test_product_does_not_import_oracle():
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import harp_amd, harp_amd.engine, harp_amd.ops, harp_amd.synth, "
            "harp_amd.manopth.manolayer; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])
    for dirpath, _, files in os.walk(os.path.join(ROOT, "harp_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().replace("# oracle", ""), f
