"""Builds libdlimgedit.so (HIP kernels + C++ host runtime + C-ABI) for gfx950 with hipcc.

In-tree, no cmake: `python -m dlimgedit_amd.build`.  hipcc cross-compiles without a GPU, so this also
runs in the CPU-only build container.  Objects are cached by source mtime under csrc/_obj/.

Three libraries come out of the same sources:
  lib/libdlimgedit.so         the product: dlimg_init + the extension entry points of include/dlimgedit/dlimgedit_amd.h,
                              named one by one in csrc/exports.map
  lib/libdlimgedit_test.so    the product's OBJECTS plus csrc/test_hooks.cpp: additionally exports the dlimg_amd_test_* /
                              dlimg_amd_bench_* hooks (include/dlimgedit/dlimgedit_amd_test.h) the parity tests and the
                              kernels-alone timing loops call; built by default next to the product
  lib/libdlimgedit_tuning.so  `--tuning`: everything compiled with -DDLIMG_TUNING -- additionally the ablated kernel variants
                              and in-kernel cycle stamps the scripts under tools/ use (DLIMGEDIT_*_ABLATE, GemmArgs::stamps);
                              tools select it with DLIMGEDIT_TUNING_LIB=1 (or the file name of a copy kept under lib/).
                              DLIMG_TUNING_DEFS in the environment adds compiler flags to this build only (A/B of variants).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
OBJ = CSRC / "_obj"
OBJ_TUNING = CSRC / "_obj_tuning"
LIB = PKG / "lib" / "libdlimgedit.so"
LIB_TEST = PKG / "lib" / "libdlimgedit_test.so"
LIB_TUNING = PKG / "lib" / "libdlimgedit_tuning.so"
ARCH = "gfx950"
SONAME = "libdlimgedit.so.1"
EXPORTS_MAP = CSRC / "exports.map"
EXPORTS_TEST_MAP = CSRC / "exports_test.map"
HOOK_SOURCES = ["test_hooks.cpp"]          # test and tuning libraries only

SOURCES = [
    "kernels/gemm.hip",
    "kernels/elementwise.hip",
    "kernels/attention_window.hip",
    "kernels/attention_global.hip",
    "kernels/decoder.hip",
    "kernels/decoder_image.hip",
    "kernels/postprocess.hip",
    "kernels/resize.hip",
    "kernels/objects.hip",
    "resize_tables.cpp",
    "image_io.cpp",
    "image_memory.cpp",
    "jpeg_decode.cpp",
    "weights.cpp",
    "sam_model.cpp",
    "environment.cpp",
    "segmentation.cpp",
    "dlimgedit.cpp",
    "ext_api.cpp",
]


# Kernels whose results must be bit-identical to the CPU oracle are compiled without mul+add contraction
# (HIP's default is -ffp-contract=fast, which also fuses the __fmul_rn/__fadd_rn header wrappers).
EXTRA_FLAGS = {
    "kernels/postprocess.hip": ["-ffp-contract=off"],
    "kernels/resize.hip": ["-ffp-contract=off"],
    "kernels/objects.hip": ["-ffp-contract=off"],
    "resize_tables.cpp": ["-ffp-contract=off"],
}


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found; the MI355X build of dlimgedit needs the ROCm toolchain")
    return exe


def _flags() -> list:
    # -fno-slp-vectorize (every source): packed fp32 arithmetic (v_pk_fma_f32 ...) only where the source asks for it with
    # vector types.  Round 3: the SLP vectoriser paired the rows of the decoder's token linears into chains of
    # v_pk_fma_f32 with op_sel, and that code -- no other -- gave one wrong element in about 10^4 decodes while several
    # lanes kept the GPU busy; the same source is clean without the pass, and nothing got slower.
    return [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DDLIMGEDIT_EXPORTS",
            "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-Wno-unused-result", f"-I{ROOT / 'include'}", f"-I{CSRC}"]


def _deps_mtime() -> float:
    hdrs = list(CSRC.rglob("*.hpp")) + list(CSRC.rglob("*.inc")) + list((ROOT / "include").rglob("*.h"))
    return max(p.stat().st_mtime for p in hdrs)


def _compile(src: str, force: bool, hdr_mtime: float, tuning: bool = False) -> Path:
    s = CSRC / src
    if not s.exists():
        raise FileNotFoundError(s)
    o = (OBJ_TUNING if tuning else OBJ) / (src.replace("/", "_") + ".o")
    if not force and o.exists() and o.stat().st_mtime > max(s.stat().st_mtime, hdr_mtime):
        return o
    cmd = [hipcc(), *_flags(), *(["-DDLIMG_TUNING", *os.environ.get("DLIMG_TUNING_DEFS", "").split()] if tuning else []), *EXTRA_FLAGS.get(src, []), "-x", "hip", "-c",
           str(s), "-o", str(o)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return o


LLVM_OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# the toolchain the -fno-slp-vectorize workaround was validated with; another compiler may pair
# fp32 operations by a different route, which is what check_packed_select_erratum() is there to catch
VALIDATED_HIPCC = "HIP 7.2.26015 / AMD clang 22.0.0git roc-7.2.0 (7b800a19)"


def disassemble_device_code(obj: Path) -> str:
    """gfx950 ISA of the device code bundled in a host object (llvm-objdump --offloading extracts next to the input,
    so it works on a copy in a scratch directory)."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        tmp = Path(d) / obj.name
        shutil.copy(obj, tmp)
        subprocess.run([LLVM_OBJDUMP, "--offloading", str(tmp)], capture_output=True, text=True, check=True)
        code = [f for f in Path(d).iterdir() if f.name.startswith(obj.name + ".") and "amdgcn" in f.name]
        if not code:
            raise RuntimeError(f"no gfx950 code object found in {obj}")
        return subprocess.run([LLVM_OBJDUMP, "-d", str(code[0])], capture_output=True, text=True, check=True).stdout


def check_packed_select_erratum(obj: Path) -> dict:
    """Build-time guard against the gfx950 packed-fp32 operand-select erratum (tools/pkfma_hazard.cpp).

    Measured on MI355X (round 4): a v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 whose LOW lane takes src1 from the HIGH
    register of the pair (op_sel:[_,1,...]) reads 0.0 for that operand about once in 1e6 executions while ANOTHER wave
    on the same SIMD has an MFMA in flight -- on settled registers, with any number of wait states around it, so it is
    not a dependency hazard and no padding cures it.  Selects on src0 or src2, op_sel_hi selects, and packed operations
    without selects never failed (0 in 5e11 each).  hipcc 7.2 emits the bad form when the SLP vectoriser pairs scalar
    FMAs of two rows (round 3: one wrong element in ~1e4 decodes); every source is therefore built with
    -fno-slp-vectorize, and this check refuses any kernel object that contains a packed instruction with a src1
    low-lane select, whoever emitted it.  Returns {kernel: number of packed fp32 instructions} for the log."""
    import re
    counts, bad, name = {}, [], None
    for line in disassemble_device_code(obj).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r"\b(v_pk_\w+)\s", line)
        if not (name and m):
            continue
        if m.group(1).endswith("_f32"):
            counts[name] = counts.get(name, 0) + 1
        sel = re.search(r"op_sel:\[([01]),([01])", line)
        if sel and sel.group(2) == "1":
            bad.append((name, line.strip().split("//")[0].strip()))
    if bad:
        listing = "\n".join(f"  {n}: {i}" for n, i in bad[:8])
        raise RuntimeError(f"{obj.name}: packed instruction with a src1 low-lane select (op_sel:[_,1,..]) -- reads 0.0 beside another "
                           f"wave's MFMA on gfx950 (DESIGN.md section 6; validated toolchain: {VALIDATED_HIPCC}):\n{listing}")
    return counts


def _link(out: Path, objs: list, exports: Path, soname: str = "") -> None:
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(out), *map(str, objs),
           "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined", *([f"-Wl,-soname,{soname}"] if soname else []),
           f"-Wl,--version-script={exports}", "-ldl", "-lz"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")


def _stale(lib: Path, objs: list, force: bool) -> bool:
    return force or not lib.exists() or lib.stat().st_mtime < max(o.stat().st_mtime for o in objs)


def build(force: bool = False, verbose: bool = False, tuning: bool = False) -> Path:
    (OBJ_TUNING if tuning else OBJ).mkdir(parents=True, exist_ok=True)
    LIB.parent.mkdir(parents=True, exist_ok=True)
    hdr = _deps_mtime()
    workers = min(6, os.cpu_count() or 1)
    with ThreadPoolExecutor(workers) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr, tuning), SOURCES))
        hooks = list(ex.map(lambda s: _compile(s, force, hdr, tuning), HOOK_SOURCES))
    if tuning:
        if _stale(LIB_TUNING, objs + hooks, force):
            _link(LIB_TUNING, objs + hooks, EXPORTS_TEST_MAP)
        if verbose:
            print(f"built {LIB_TUNING} ({LIB_TUNING.stat().st_size / 1e6:.1f} MB)")
        return LIB_TUNING
    if _stale(LIB, objs, force) or _stale(LIB_TEST, objs + hooks, force):
        for o in objs:                                          # before anything is linked: a refused build leaves no library
            if o.name.startswith("kernels_"):
                check_packed_select_erratum(o)
        if _stale(LIB, objs, force):
            _link(LIB, objs, EXPORTS_MAP, SONAME)
        # the test library: the SAME objects (what the parity tests exercise is the product's code) + the hooks
        _link(LIB_TEST, objs + hooks, EXPORTS_TEST_MAP)
    # consumers linked against the reference's library resolve libdlimgedit.so.1 (SOVERSION 1,
    # /root/reference/src/CMakeLists.txt:20-23)
    link = LIB.parent / SONAME
    if not link.is_symlink() and not link.exists():
        link.symlink_to(LIB.name)
    if verbose:
        print(f"built {LIB} ({LIB.stat().st_size / 1e6:.1f} MB) and {LIB_TEST.name}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, tuning="--tuning" in sys.argv)
