"""Build libcreste_hip.so (gfx950) in-tree with hipcc.

    python -m creste_public_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to creste_public_amd/lib/obj, the shared library
to creste_public_amd/lib/libcreste_hip.so (git-ignored; it travels to the GPU box with the repo
snapshot).  `-ffp-contract=off`: the geometry / value-iteration kernels spell out every fma they
want, so results do not depend on the compiler's contraction choices.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libcreste_hip.so")
ARCH = "gfx950"
SOURCES = ["error.cpp", "plan_runtime.cpp", "conv_igemm.hip", "conv_patch.hip", "conv_wino.hip", "conv_wino4.hip", "pointwise.hip", "bev_splat.hip", "value_iteration.hip",
           "svf.hip", "planner.hip", "lidar.hip", "train.hip", "train_backbone.hip", "losses.hip", "labels.hip", "mbconv.hip", "supcon_mfma.hip", "conv_chain.hip"]
# per-source extra flags.  conv_wino.hip: no SLP vectorisation -- packed fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32) costs a
# wave ~12 cycles per instruction next to the partner wave's MFMAs (MI355X_MICROARCH.md), the scalar forms do not
EXTRA_FLAGS = {"conv_wino.hip": ["-fno-slp-vectorize"], "conv_wino4.hip": ["-fno-slp-vectorize"]}
if os.environ.get("CRESTE_W4_EXPERIMENTS") == "1":          # timing-experiment instantiations of the F(4x4) GEMM (scripts/w4_gemm_exp.sh)
    EXTRA_FLAGS["conv_wino4.hip"] = EXTRA_FLAGS["conv_wino4.hip"] + ["-DCRESTE_W4_EXPERIMENTS"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}",
         "-Wall", "-Wno-unused-function"]
# NO packed-fp32 VALU anywhere (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): on gfx950 a wave's packed-fp32 result is
# occasionally WRONG -- one component of 16 consecutive lanes of one instruction -- while waves of ANOTHER kernel issue MFMAs
# on the same CU (found when the inference forward started to run two half-batches on two streams: the fused MBConv kernels
# beside the 1x1 conv kernel, 3-8 corrupted steps of 12; none of 120 with scalar v_fma_f32; scripts/concurrency_bisect.py,
# profiles/r04_pipeline_notes.md).  One stream never shows it (kernels of one stream do not share a CU), any second stream
# can (pipelined inference, the IRL step's prefetch of the frozen half).  Speed: neutral (39.8 -> 39.4 ms for the step).
NO_PK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS += NO_PK


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(HERE, "..", "include", "creste_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_m = _deps_mtime()
    objs, rebuilt = [], False
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_m):
            cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c", sp, "-o", op]
            if verbose:
                print("[creste build]", " ".join(cmd), flush=True)
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            remarks = [ln for ln in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" in ln]
            other = [ln for ln in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in ln
                     and "is not a recognized feature for this target" not in ln]     # (the host half ignores NO_PK)
            if r.returncode or any("warning:" in ln or "error:" in ln for ln in other):   # (else: the remarks' source context)
                print("\n".join(other), file=sys.stderr, flush=True)
            if r.returncode:
                raise subprocess.CalledProcessError(r.returncode, cmd)
            with open(op + ".usage.txt", "w") as f:        # registers / scratch / occupancy of every kernel: resource_usage()
                f.write("\n".join(remarks) + "\n")
            rebuilt = True
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print("[creste build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


def build_variant(tag: str, pk_sources=(), verbose: bool = True, extra_flags=()) -> str:
    """An EXPERIMENT library lib/libcreste_hip_<tag>.so: the same sources, `pk_sources` compiled WITH packed-fp32 VALU allowed
    (scripts/pk_hazard.sh: is the corruption of round 4 still there, and in which file).  Load it with CRESTE_HIP_LIB=<path>.
    Never the shipped library: tests/test_abi.py disassembles lib/libcreste_hip.so only."""
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, "obj_" + tag)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        op = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        if src in pk_sources:
            op = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
            flags = [f for f in FLAGS if f not in NO_PK]
            cmd = [hipcc, *flags, *EXTRA_FLAGS.get(src, []), *extra_flags, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", op]
            if verbose:
                print("[creste build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(op)
    lib = os.path.join(LIBDIR, f"libcreste_hip_{tag}.so")
    subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", lib], check=True)
    return lib


def resource_usage() -> dict:
    """{mangled kernel name: {"vgprs", "agprs", "scratch", "occupancy", "lds", "source"}} of the last build (from the
    compiler's kernel-resource-usage remarks; tests/test_abi.py keeps the hot kernels out of scratch with it)."""
    import re
    out = {}
    for src in SOURCES:
        path = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o.usage.txt")
        if not os.path.exists(path):
            continue
        cur = None
        for ln in open(path):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                cur = out.setdefault(m.group(1), {"source": src})
                continue
            if cur is None:
                continue
            for key, pat in (("vgprs", r"VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, ln)
                if m:
                    cur[key] = int(m.group(1))
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:                    # python -m creste_public_amd.build --variant pk mbconv.hip [more.hip ...]
        i = sys.argv.index("--variant")
        build()
        print(build_variant(sys.argv[i + 1], tuple(sys.argv[i + 2:])))
    else:
        print(build(force="--force" in sys.argv))
