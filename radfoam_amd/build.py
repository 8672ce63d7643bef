"""Builds libradfoam_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(_HERE, "csrc", "rf_kernels.hip"), os.path.join(_HERE, "csrc", "rf_scene_ops.hip")]
HEADERS = [os.path.join(_HERE, "csrc", "rf_math.hpp"), os.path.join(_HERE, "csrc", "rf_foam.hpp"),
           os.path.join(_HERE, "csrc", "rf_wave.hpp"), os.path.join(_HERE, "csrc", "rf_host.hpp"),
           os.path.join(os.path.dirname(_HERE), "include", "radfoam_hip.h")]
OUTPUT = os.path.join(_HERE, "libradfoam_hip.so")

# -ffp-contract=off: the kernels spell out every FMA (csrc/rf_math.hpp); nothing else may fuse.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-Wno-unused-result"]


def needs_build() -> bool:
    if not os.path.exists(OUTPUT):
        return True
    out_m = os.path.getmtime(OUTPUT)
    return any(os.path.getmtime(p) > out_m for p in SOURCES + HEADERS)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUTPUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", OUTPUT] + SOURCES
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(cmd))
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed building libradfoam_hip.so")
    return OUTPUT


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
