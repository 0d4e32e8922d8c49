"""Builds libpie_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m proxy_inference_engine_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored and travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB = LIB_DIR / "libpie_hip.so"
SOURCES = ["w4_gemv.hip", "ops.hip", "decoder.hip", "prefill.hip", "vision.hip", "w4m_gemm.hip", "tp_comm.hip", "sampler.hip", "paged_i8.hip", "page_pool.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.cpp")) + [PKG.parent / "include" / "pie_hip.h"]
    return any(d.stat().st_mtime > t for d in deps)


def source_hash() -> str:
    """sha256 over the sources the library is built from: pie_version() carries its first 12 digits, and the counter files under
    profiles/ name the build they were measured on (bench.py refuses a stale one)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.cpp")) + [PKG.parent / "include" / "pie_hip.h"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:12]


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not _stale():
        return LIB
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = LIB_DIR / "obj"
    obj_dir.mkdir(exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: str) -> Path:
        obj = obj_dir / (src + ".o")
        extra = [f'-DPIE_BUILD_HASH="{source_hash()}"'] if src == "decoder.hip" else []
        cmd = [hipcc, *FLAGS, *extra, "-c", str(CSRC / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-6000:]}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs), "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
