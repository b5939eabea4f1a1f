"""Build libv2p_rollout.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the repo snapshot.  `python -m vid2player3d_amd.build`
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libv2p_rollout.so")
SOURCES = ["capi.hip", "motion_state.hip", "task_ops.hip", "shape_compile.hip", "motion_build.hip", "physics.hip", "physics_ll.hip"]
HEADERS = ["v2p_internal.hpp", "v2p_dev.hpp", "v2p_math.inc", "phys_math.hpp", "phys_common.hpp", "motion_sample.inc", "hull_gjk.hpp", "post_ops.inc", os.path.join("..", "..", "include", "v2p_rollout.h")]
ARCH = "gfx950"


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_info():
    """What the bench line records about the build."""
    return {"arch": ARCH, "kernel_source_sha16": kernel_source_hash(), "physics_ll_math": "precise" if os.environ.get("V2P_LL_STRICT_MATH") else "relaxed (" + " ".join(LL_MATH_FLAGS) + "; precise device libraries, strict pre-physics prologue)",
            "hipcc": _hipcc()}


LL_CODEGEN_FLAGS = ["-mllvm", "-sink-insts-to-avoid-spills=1"]
LL_MATH_FLAGS = ["-fassociative-math", "-freciprocal-math", "-fno-signed-zeros", "-fno-trapping-math", "-fno-honor-nans"]
# what the physics kernel is compiled from: the counters kept under profiles/ (VALU instructions, HBM bytes per launch) describe ONE
# kernel; they carry this hash, and bench.py drops them from its line when the sources have moved on
KERNEL_SOURCES = ["physics_ll.hip", "phys_common.hpp", "phys_math.hpp", "hull_gjk.hpp", "post_ops.inc", "v2p_math.inc", "motion_sample.inc", "v2p_internal.hpp", "v2p_dev.hpp"]


def kernel_source_hash():
    """sha256 (16 hex digits) of the sources of physics_ll_kernel and of the flags it is built with."""
    import hashlib

    h = hashlib.sha256()
    for s in KERNEL_SOURCES:
        with open(os.path.join(CSRC, s), "rb") as f:
            h.update(s.encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(LL_MATH_FLAGS + LL_CODEGEN_FLAGS + [os.environ.get("V2P_LL_STRICT_MATH", ""), os.environ.get("V2P_FLAGS_PHYSICS_LL", "")]).encode())
    return h.hexdigest()[:16]


def build(force=False, verbose=False, lib_out=None, tag=""):
    """out / tag: build a variant library next to the default one (A/B experiments: tools/variants.sh)."""
    if lib_out is None and not force and not needs_build():
        return LIB
    objs = []
    # -fno-slp-vectorize: the SLP vectoriser packs adjacent fp32 ops into v_pk_* on 64-bit register pairs, which costs the
    # physics kernels ~160 registers of pressure (1 instead of 2 waves/SIMD) and is slower next to dependent chains anyway
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-vectorize", "-fno-slp-vectorize", "-Wall",
             "-Wno-unused-function"]
    procs = []
    # (physics_ll.hip is compiled twice: the default object and the register build, in its own namespace - see csrc/capi.hip)
    for s in SOURCES + ["physics_ll.hip:regs"]:
        regs = s.endswith(":regs")
        s = s.split(":")[0]
        obj = os.path.join(CSRC, s.replace(".hip", ("_regs" if regs else "") + tag + ".o"))
        fl = list(flags)
        if regs:
            # (the ILP scheduler: this build runs where a launch is as long as its heaviest wave's chain, +1.6 % at 4096 envs, +2.1 % at 1024;
            # for the default build, bound by instruction issue at three waves per SIMD, the same switch costs 0.3 %: profiles/r04e_dual_build.txt)
            fl += ["-Dv2p=v2p_regs", "-DV2P_LL_WPS=2", "-DV2P_LL_WPS_BALL=2", "-DV2P_LL_WPS_LIMITS=2", "-DV2P_LL_PARK2=0", "-DV2P_LL_PARK3=0",
                   "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
        if s in ("motion_state.hip", "task_ops.hip", "shape_compile.hip", "motion_build.hip"):
            # the task-side kernels restate torch elementwise code: no FMA contraction, so that ill-conditioned spots of the
            # reference itself (acos of a dot product next to 1 in slerp / angle-axis) round the way torch rounds them
            fl = [f if f != "-ffp-contract=fast" else "-ffp-contract=off" for f in fl]
        if s == "physics_ll.hip":
            # the link-per-lane physics kernel: relaxed fp32 arithmetic for the code of this file, precise device libraries (see the head
            # of the file for why the flags are spelled out instead of -ffast-math); V2P_LL_STRICT_MATH=1 builds it precise (A/B, bisecting)
            fl = [f if f != "-ffp-contract=fast" else "-ffp-contract=fast-honor-pragmas" for f in fl]
            fl = fl + (["-DV2P_LL_STRICT_MATH"] if os.environ.get("V2P_LL_STRICT_MATH") else LL_MATH_FLAGS)
            # MachineLICM hoists every loop-invariant address / uniform expression in front of the substep loop and the register allocator then
            # spills them across it; this switch lets it sink them back instead (round 6: racket + ball + limits 328 -> 184 B of scratch per
            # lane, 12.89 -> 13.64 M env-steps/s; limits alone 84 -> 0 B; the headline instantiation 20 -> 0 B, unchanged speed:
            # profiles/r06d_variants_spill.log)
            fl = fl + LL_CODEGEN_FLAGS
        extra = os.environ.get("V2P_FLAGS_" + s.split(".")[0].upper() + ("_REGS" if regs else ""))  # experiments: per-object flag override, e.g. V2P_FLAGS_PHYSICS_LL="-O1" (V2P_FLAGS_PHYSICS_LL_REGS: the register build)
        if extra:
            fl = [f for f in fl if f != "-O3"] + extra.split()
        cmd = [_hipcc()] + fl + ["-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        if verbose and out:
            print(out.decode())
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib_out or LIB] + objs
    subprocess.check_call(cmd)
    return lib_out or LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
