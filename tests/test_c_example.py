"""The C ABI from plain C (examples/raster_c_abi.c): gcc compiles it against include/harp_hip.h as C11 — no HIP compiler, no C++, no torch —
and links it with libharp_hip.so; on the GPU box the binary runs and checks the rasteriser's output itself."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "raster_c_abi.c")
LIBDIR = os.path.join(ROOT, "harp_amd", "csrc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(out):
    from harp_amd import build
    build.build(force=False, verbose=False)
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROCM, "include"), SRC,
           "-L", LIBDIR, "-lharp_hip", "-L", os.path.join(ROCM, "lib"), "-lamdhip64", "-lm", f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{ROCM}/lib", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_example_compiles_and_links_as_c11(tmp_path):
    exe = _build(str(tmp_path / "raster_c_abi"))
    # every harp_* symbol the example calls is resolved from libharp_hip.so (undefined in the executable, defined in the library)
    und = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    assert "harp_rasterize_fwd" in und and "harp_rasterize_ws_bytes" in und


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    exe = _build(str(tmp_path / "raster_c_abi"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok")
