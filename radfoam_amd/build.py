"""Builds libradfoam_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every source is compiled to its own object (in parallel, rebuilt only when it or a header is newer)
and the objects are linked into the shared library; the A/B scripts under scripts/ compile
variants of single sources the same way.
"""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
SOURCES = [os.path.join(_CSRC, n) for n in ("rf_kernels.hip", "rf_scene_ops.hip", "rf_adjacency.hip", "rf_grad_exchange.hip", "rf_delaunay.hip", "rf_tile_prior.hip")]
HEADERS = [os.path.join(_CSRC, n) for n in ("rf_math.hpp", "rf_foam.hpp", "rf_wave.hpp", "rf_host.hpp", "rf_star.hpp", "rf_tiles.hpp")] + [
    os.path.join(os.path.dirname(_HERE), "include", "radfoam_hip.h")]
OBJ_DIR = os.path.join(_CSRC, "_obj")
OUTPUT = os.path.join(_HERE, "libradfoam_hip.so")

# -ffp-contract=off: the kernels spell out every FMA (csrc/rf_math.hpp); nothing else may fuse.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-fno-fast-math", "-Wno-unused-result"]


def _hipcc() -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def _object(src: str) -> str:
    return os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash() -> str:
    """sha256 over the kernel sources and headers, in a fixed order: what a set of hardware counters was measured on
    (profiles/counters.json records it; bench.py refuses to quote counters of another build)."""
    import hashlib

    h = hashlib.sha256()
    for path in sorted(SOURCES + HEADERS):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build() -> bool:
    return _stale(OUTPUT, SOURCES + HEADERS)


def _run(cmd, verbose):
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(cmd))
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed building libradfoam_hip.so")


def build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUTPUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = [s for s in SOURCES if force or _stale(_object(s), [s] + HEADERS)]
    with ThreadPoolExecutor(max_workers=max(1, len(todo))) as pool:
        list(pool.map(lambda s: _run([_hipcc()] + HIPCC_FLAGS + ["-c", s, "-o", _object(s)], verbose), todo))
    _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUTPUT] + [_object(s) for s in SOURCES],
         verbose)
    return OUTPUT


if __name__ == "__main__":
    import sys

    if "--source-hash" in sys.argv:
        print(source_hash())
    else:
        print(build_hip(force=True, verbose=True))
