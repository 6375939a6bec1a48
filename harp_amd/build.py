"""Build the C-ABI shared library (harp_amd/csrc/libharp_hip.so) for gfx950 with hipcc.

    python -m harp_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the tree.
"""
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libharp_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
         "-Wno-unused-result", "-DNDEBUG"] + os.environ.get("HARP_EXTRA_FLAGS", "").split()


# per-file extra flags.  shade_bwd.hip: the SLP vectoriser pairs float operations into v_pk_* instructions, whose aligned register
# pairs (and the moves that build them) cost the shader backward ~15 VGPRs; that kernel is bound by waves in flight, not by VALU issue
# (155 -> 122 VGPRs together with the reload of the face data in its backward half: 3 -> 4 waves per SIMD, no scratch)
# ... and the "max-memory-clause" machine scheduler (loads grouped into clauses ahead of their uses) instead of the default: same registers, same
# results, step -2.4 us in 8 of 8 pairs of runs (profiles/r06_ab_record.txt 26; "max-ilp": shader backward 0.190 -> 0.214 ms)
# conv.hip: the same scheduler, float32 perceptual term 30.53 30.54 30.53 -> 30.10 30.04 30.09 ms (bf16 split unchanged); the other files: no effect
_MEM_CLAUSE = ["-mllvm", "--amdgpu-sched-strategy=max-memory-clause"]
# raster.hip: without the SLP vectoriser as well — the three tile kernels are VALU-issue-bound and register-tight (camera view 72 VGPRs with 12 B
# of scratch): light view 58 -> 52 VGPRs, silhouette backward's scratch gone; camera group 0.164 -> 0.160 ms, light 0.086 -> 0.082, step -10 us (4 of 4)
FILE_FLAGS = {"shade_bwd.hip": ["-fno-slp-vectorize"] + _MEM_CLAUSE, "conv.hip": _MEM_CLAUSE, "raster.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(CSRC, "..", "..", "include", "harp_hip.h"), os.path.abspath(__file__)]      # (this file: the flags)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in sources():
        o = s[:-4] + ".o"
        objs.append(o)
        cmd = [hipcc, "-c", *[f for f in FLAGS if f != "-shared"], *FILE_FLAGS.get(os.path.basename(s), []), "-I",
               os.path.join(CSRC, "..", "..", "include"), s, "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", LIB])
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
