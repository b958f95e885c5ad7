"""Drop-in boundary checks that need no GPU: the library loads, exports what include/*.h declares,
the POD layouts match the reference's, and the non-throwing entry points behave."""
import ctypes as C
import re
import shutil
import subprocess
import sys
import threading
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
INCLUDE = ROOT / "include"


@pytest.fixture(scope="module")
def api():
    from dlimgedit_amd.build import build
    build()                                   # hipcc cross-compiles without a GPU
    from dlimgedit_amd import api
    return api


def _declared(header: Path) -> set:
    return set(re.findall(r"DLIMG_API\s+[\w\s\*]+?\b(dlimg_\w+)\s*\(", header.read_text()))


def test_library_exports_every_declared_symbol(api):
    """Product library <-> dlimgedit.h + dlimgedit_amd.h; test library <-> dlimgedit_amd_test.h on top of those."""
    product = _declared(INCLUDE / "dlimgedit" / "dlimgedit.h") | _declared(INCLUDE / "dlimgedit" / "dlimgedit_amd.h")
    hooks = _declared(INCLUDE / "dlimgedit" / "dlimgedit_amd_test.h")
    every = set()
    for header in INCLUDE.rglob("*.h"):
        every |= _declared(header)
    assert every == product | hooks, "a header declares an entry point that no library is checked for"
    assert "dlimg_init" in product and len(product) >= 15
    assert product == {"dlimg_init", *api.ext.EXPORTS}, "python binding and dlimgedit_amd.h disagree"
    assert hooks == set(api.ext.HOOK_EXPORTS), "python binding and dlimgedit_amd_test.h disagree"
    lib, hooks_lib = api.library(), api.hooks_library()
    for name in product:
        assert getattr(lib, name) is not None
    for name in product | hooks:
        assert getattr(hooks_lib, name) is not None


def _dynamic_symbols(lib: Path) -> list:
    out = subprocess.run(["nm", "-D", "--defined-only", str(lib)], capture_output=True, text=True, check=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if l.strip())


def test_only_dlimg_symbols_are_exported(api):
    """The reference hides everything but dlimg_init (/root/reference/src/CMakeLists.txt:11, dlimgedit.h:70).  The PRODUCT
    library: dlimg_init + the 23 extension entry points of dlimgedit_amd.h, named one by one in csrc/exports.map, and NOTHING
    else of any symbol type -- no test or benchmark hook, no weak libstdc++ template instantiation
    (std::filesystem::path::..., std::vector<...>::~vector).  The hooks live in the test library only."""
    lib_dir = ROOT / "dlimgedit_amd" / "lib"
    names = _dynamic_symbols(lib_dir / "libdlimgedit.so")
    assert names == sorted({"dlimg_init", *api.ext.EXPORTS}), sorted(set(names) ^ {"dlimg_init", *api.ext.EXPORTS})
    assert len(names) == 24 and not [n for n in names if "_test_" in n or "_bench_" in n]
    test_names = _dynamic_symbols(lib_dir / "libdlimgedit_test.so")
    want = sorted({"dlimg_init", *api.ext.EXPORTS, *api.ext.HOOK_EXPORTS})
    assert test_names == want, sorted(set(test_names) ^ set(want))
    assert len(api.ext.HOOK_EXPORTS) == 20 and all("_test_" in n or "_bench_" in n for n in api.ext.HOOK_EXPORTS)


def test_hooks_come_from_the_test_library_not_the_product(api):
    """A hook call loads lib/libdlimgedit_test.so (the product's objects + csrc/test_hooks.cpp); the product library has no
    such symbol to fall back on, and a hook's error text comes from the library it ran in."""
    assert api.hooks_library() is not api.library()
    assert not hasattr(api.library(), "dlimg_amd_test_plan_steps")
    passes, *_ = api.ext.plan_steps([0, 0], [0, 0], 0, 3, 2, 4, True)
    assert sum(n for _, n in passes) == 3
    with pytest.raises(api.Error, match="piece array is too small|assert|Assertion"):
        api.ext._h().dlimg_amd_test_mask_pieces.argtypes        # (signatures are set)
        sizes = (C.c_longlong * 1)(1 << 40)
        got = api.ext._h().dlimg_amd_test_mask_pieces(1, sizes, 0, (C.c_longlong * 1)(), 0, (C.c_longlong * 5)(), 1, C.byref(C.c_int()))
        assert got < 0
        api._check_hook(1)


def test_pod_layouts_match_reference(api):
    """dlimg_ImageView 24 B (pixels@16), dlimg_Options 16 B (model_directory@8), 13 reference slots."""
    assert C.sizeof(api._ImageView) == 24 and api._ImageView.pixels.offset == 16
    assert api._ImageView.stride.offset == 12 and api._ImageView.channels.offset == 8
    assert C.sizeof(api._Options) == 16 and api._Options.model_directory.offset == 8
    names = [n for n, _ in api._API_FIELDS]
    assert names[:13] == ["is_backend_supported", "create_environment", "destroy_environment",
                          "process_image_for_segmentation", "get_segmentation_mask", "get_segmentation_extent",
                          "destroy_segmentation", "segment_objects", "load_image", "save_image", "create_image",
                          "destroy_image", "last_error"]
    assert api.REFERENCE_SLOTS == 13 and C.sizeof(api._Api) == 8 * len(names)
    table = api.api()
    for n in names:
        assert getattr(table, n), f"slot {n} is null"


def test_header_is_valid_c_and_cpp(tmp_path):
    cc = shutil.which("gcc")
    src = tmp_path / "t.c"
    src.write_text('#include <dlimgedit/dlimgedit_amd_test.h>\n'
                   '_Static_assert(sizeof(dlimg_ImageView) == 24, "view");\n'
                   '_Static_assert(sizeof(dlimg_Options) == 16, "options");\n'
                   '_Static_assert(sizeof(dlimg_Api) == 15 * sizeof(void*), "table");\n'
                   'int main(void) { dlimg_Api const* a = 0; (void)a; return 0; }\n')
    subprocess.run([cc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", f"-I{INCLUDE}", str(src)], check=True)
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include <dlimgedit/dlimgedit.hpp>\nint main() { return dlimg::count(dlimg::Channels::bgra) == 4 ? 0 : 1; }\n')
    subprocess.run([shutil.which("g++"), "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", f"-I{INCLUDE}", str(cpp)],
                   check=True)


def test_init_is_idempotent(api):
    lib = api.library()
    assert C.addressof(lib.dlimg_init().contents) == C.addressof(lib.dlimg_init().contents)


def test_backend_probe_never_throws(api):
    assert api.Environment.is_supported(api.Backend.cpu) is False     # no CPU execution path in this build
    assert api.Environment.is_supported(api.Backend.gpu) in (True, False)
    assert api.ext.device_count() >= 0


def test_create_and_destroy_image(api):
    img = api.Image(api.Extent(8, 6), api.Channels.bgra)
    assert img.size() == 8 * 6 * 4 and img.pixels().shape == (6, 8, 4)
    img.pixels()[:] = 9
    assert api.api().create_image(0, 4, 4) is None


def test_environment_errors_are_reported_through_last_error(api, tmp_path):
    with pytest.raises(api.Error, match="does not exist"):
        api.Environment(api.Options(api.Backend.gpu, str(tmp_path / "missing")))
    f = tmp_path / "file"
    f.write_text("x")
    with pytest.raises(api.Error, match="is not a directory"):
        api.Environment(api.Options(api.Backend.gpu, str(f)))
    with pytest.raises(api.Error, match="CPU backend is not available"):
        api.Environment(api.Options(api.Backend.cpu, str(tmp_path)))


def test_cpu_requests_can_be_sent_to_the_gpu_by_the_deployer(api, tmp_path, monkeypatch):
    """The reference's default Options ask for Backend::cpu (/root/reference/src/include/dlimgedit/dlimgedit.hpp:91); this build
    has no CPU path and refuses -- unless the deployer, who cannot recompile the consumer, sets DLIMGEDIT_CPU_REQUESTS_ON_GPU=1:
    then such a request is a GPU request (here, without a GPU, it gets the GPU path's own answer, not the CPU refusal)."""
    monkeypatch.setenv("DLIMGEDIT_CPU_REQUESTS_ON_GPU", "1")
    assert api.Environment.is_supported(api.Backend.cpu) == api.Environment.is_supported(api.Backend.gpu)
    if not api.Environment.is_supported(api.Backend.gpu):
        with pytest.raises(api.Error, match="No supported GPU"):
            api.Environment(api.Options(api.Backend.cpu, str(tmp_path)))
    monkeypatch.setenv("DLIMGEDIT_CPU_REQUESTS_ON_GPU", "0")
    assert api.Environment.is_supported(api.Backend.cpu) is False
    with pytest.raises(api.Error, match="CPU backend is not available"):
        api.Environment(api.Options(api.Backend.cpu, str(tmp_path)))


def test_image_file_slots_work_without_a_gpu(api, tmp_path):
    """load_image / save_image are host code (PNG; tests/test_image_io.py has the details)."""
    import numpy as np
    with pytest.raises(api.Error, match="Failed to load image nothing.png"):
        api.Image.load("nothing.png")
    px = np.arange(2 * 2 * 4, dtype=np.uint8).reshape(2, 2, 4)
    api.Image.save(api.ImageView(px), tmp_path / "x.png")
    assert np.array_equal(api.Image.load(tmp_path / "x.png").pixels(), px)


def test_last_error_is_per_thread(api, tmp_path):
    """The reference keeps one unsynchronised global string; here each thread sees its own message."""
    seen = {}

    def worker(name):
        try:
            api.Environment(api.Options(api.Backend.gpu, str(tmp_path / name)))
        except api.Error as e:
            seen[name] = str(e)

    ts = [threading.Thread(target=worker, args=(f"dir{i}",)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(4):
        assert f"dir{i}" in seen[f"dir{i}"]


def test_missing_gpu_fails_loudly_not_silently(api, tmp_path):
    """On a box without an MI355X the product path must raise; there is no CPU fallback."""
    if api.Environment.is_supported(api.Backend.gpu):
        pytest.skip("GPU present")
    (tmp_path / "segmentation").mkdir()
    with pytest.raises(api.Error, match="No supported GPU"):
        api.Environment(api.Options(api.Backend.gpu, str(tmp_path)))


def test_reference_wrapper_consumer_runs_against_this_library():
    """A program compiled against the REFERENCE's header-only C++ wrapper drives this library."""
    sys.path.insert(0, str(ROOT))
    from oracle.build_ref import build_abi_consumer
    exe = build_abi_consumer()
    if exe is None:
        pytest.skip("reference source tree not available")
    r = subprocess.run([str(exe), str(ROOT / "dlimgedit_amd" / "lib" / "libdlimgedit.so"), "probe"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "cpu=0" in r.stdout and "image size=192" in r.stdout
    assert "error=Model path /definitely/not/here does not exist" in r.stdout


def test_packed_select_guard_passes_on_the_product_and_fires_on_the_known_bad_form(api, tmp_path):
    """dlimgedit_amd/build.py::check_packed_select_erratum: no kernel may contain a packed instruction whose low lane takes
    src1 from the high register (op_sel:[_,1,..]) -- on gfx950 that operand reads 0.0 about once in 1e6 executions beside
    another wave's MFMA (tools/pkfma_hazard.cpp).  Every product kernel object passes; the known-bad
    form of the token linears (-DDLIMG_STRAIGHT_ROWS with the SLP vectoriser left on, what round 3 shipped for a day) is
    refused."""
    from dlimgedit_amd import build as B
    objs = sorted(B.OBJ.glob("kernels_*.o"))
    assert len(objs) >= 9
    for o in objs:
        B.check_packed_select_erratum(o)                   # raising is the failure mode
    bad = tmp_path / "kernels_decoder.hip.o"
    flags = [f for f in B._flags() if f != "-fno-slp-vectorize"]
    subprocess.run([B.hipcc(), *flags, "-DDLIMG_TUNING", "-DDLIMG_STRAIGHT_ROWS", "-x", "hip", "-c",
                    str(B.CSRC / "kernels" / "decoder.hip"), "-o", str(bad)], check=True, capture_output=True)
    with pytest.raises(RuntimeError, match="src1 low-lane select"):
        B.check_packed_select_erratum(bad)


def test_cpu_list_parser_of_the_numa_binding(api):
    """bind_thread_near_device (csrc/environment.cpp) reads /sys/devices/system/node/nodeN/cpulist: ranges, singles, junk."""
    assert api.ext.parse_cpu_list("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    assert api.ext.parse_cpu_list("64-67,192-193\n") == [64, 65, 66, 67, 192, 193]
    assert api.ext.parse_cpu_list("") == [] and api.ext.parse_cpu_list("x,5,7-6,-2,9") == [5, 9]
