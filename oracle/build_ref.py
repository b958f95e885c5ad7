"""Builds checker artefacts that need the upstream source tree, into oracle/_ref/ (git-ignored).

The reference's Segment-Anything path itself cannot be built here (it needs onnxruntime, Eigen, stb,
fmt and downloaded .onnx graphs, none of which exist in the image: SURVEY.md §8c).  What CAN be built
from the reference's own files is a consumer of its header-only public C++ wrapper
(src/include/dlimgedit/*.hpp, dependency-free): compiled against those headers and pointed at this
repository's libdlimgedit.so, it proves the C-ABI is a drop-in (tests/test_abi.py, tests/test_gpu_e2e.py).
Only compiler output is written; no reference source is copied.
"""
from __future__ import annotations

import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
REF_INCLUDE = Path("/root/reference/src/include")
OUT = Path(__file__).resolve().parent / "_ref"


def build_abi_consumer() -> Path | None:
    """Returns the binary path, or None when the reference tree is not present (GPU box)."""
    exe = OUT / "abi_consumer"
    src = ROOT / "tests" / "abi_consumer.cpp"
    if not (REF_INCLUDE / "dlimgedit" / "dlimgedit.hpp").exists():
        return exe if exe.exists() else None
    OUT.mkdir(parents=True, exist_ok=True)
    if exe.exists() and exe.stat().st_mtime > src.stat().st_mtime:
        return exe
    cxx = shutil.which("g++") or "g++"
    cmd = [cxx, "-std=c++17", "-O1", f"-I{REF_INCLUDE}", str(src), "-o", str(exe), "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building abi_consumer failed:\n{r.stderr}")
    return exe


if __name__ == "__main__":
    print(build_abi_consumer())
