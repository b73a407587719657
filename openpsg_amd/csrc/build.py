"""Builds libpsg_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m openpsg_amd.csrc.build [--force] [--save-temps]

The shared object is written next to the package (openpsg_amd/libpsg_hip.so) so that it travels
with the source tree; it is git-ignored.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, "libpsg_hip.so")
SOURCES = ["psg_core.hip", "psg_rowops.hip", "psg_attn.hip", "psg_attn_f32.hip", "psg_xattn_mfma.hip", "psg_xattn_dma.hip", "psg_gemm.hip", "psg_gemm_f32.hip", "psg_batch_gemm.hip", "psg_decode_layer.hip", "psg_split.hip", "psg_dense_gemm.hip", "psg_patch_embed.hip", "psg_selfattn_mfma.hip", "psg_prefill_attn_mfma.hip", "psg_pool.hip", "psg_train.hip", "psg_train_bwd.hip"]
HEADERS = ["psg_common.h", "psg_decode_math.h", os.path.join("..", "..", "include", "psg_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES + HEADERS if os.path.exists(os.path.join(HERE, s))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> str:
    """One object per source (compiled in parallel, kept under csrc/_obj/ and reused while the source and the
    headers are older), then one link: a change to one kernel file costs one compile, not twelve."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(HERE, "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
    if save_temps:
        os.makedirs(os.path.join(HERE, "_temps"), exist_ok=True)
        flags += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    hdr_t = max(os.path.getmtime(os.path.join(HERE, h)) for h in HEADERS)
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(obj_dir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or save_temps or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([_hipcc()] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=HERE)
    with ThreadPoolExecutor(max_workers=min(8, max(1, (os.cpu_count() or 2) - 1))) as ex:
        list(ex.map(run, jobs))
    run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--save-temps", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, save_temps=a.save_temps))
