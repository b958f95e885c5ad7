"""Python host binding over the C-ABI of libdlimgedit.so (ctypes, no torch types anywhere).

Mirrors the reference's header-only C++ wrapper (reference: src/include/dlimgedit/dlimgedit.hpp and
detail/dlimgedit.impl.hpp): same class and method names, same argument meaning, errors surface as
`Error(last_error())` exactly where the C++ wrapper throws `dlimg::Exception`.  Only the slots of
`dlimg_Api` are used for the drop-in surface; the `ext` namespace exposes the extension entry points
of include/dlimgedit/dlimgedit_amd.h for benchmarks and parity tests.

The library is the product: if it is missing or the GPU is absent, calls fail loudly; there is no
fallback implementation in Python.
"""
from __future__ import annotations

import ctypes as C
import os
import enum
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Sequence, Tuple

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libdlimgedit.so"
# The test and benchmark hooks (include/dlimgedit/dlimgedit_amd_test.h: ext.test_*, ext.bench_*, ext.force_gemm_tile,
# ext.mask_pieces, ext.plan_steps) are not in the product library: they come from lib/libdlimgedit_test.so, the product's own
# objects plus csrc/test_hooks.cpp, loaded next to the product the first time a hook is called.
HOOKS_LIB_PATH = LIB_PATH.with_name("libdlimgedit_test.so")
# tools/ only: the -DDLIMG_TUNING build with ablated kernel variants and in-kernel stamps (python -m dlimgedit_amd.build --tuning);
# it exports the hooks too, so one library serves both roles
if os.environ.get("DLIMGEDIT_TUNING_LIB") == "1":
    LIB_PATH = HOOKS_LIB_PATH = LIB_PATH.with_name("libdlimgedit_tuning.so")
elif os.environ.get("DLIMGEDIT_TUNING_LIB"):          # a copy of a tuning build kept under another name (A/B of several variants)
    LIB_PATH = HOOKS_LIB_PATH = LIB_PATH.with_name(os.environ["DLIMGEDIT_TUNING_LIB"])


class Error(RuntimeError):
    """dlimg::Exception."""


class Backend(enum.IntEnum):
    cpu = 0
    gpu = 1


class Channels(enum.IntEnum):
    mask = 1
    rgb = 3
    rgba = 4
    bgra = 5
    argb = 6


def count(channels: Channels) -> int:
    return 4 if int(channels) > 4 else int(channels)


@dataclass(frozen=True)
class Extent:
    width: int = 0
    height: int = 0


@dataclass(frozen=True)
class Point:
    x: int = 0
    y: int = 0


@dataclass(frozen=True)
class Region:
    top_left: Point = Point()
    bottom_right: Point = Point()

    @staticmethod
    def from_origin(origin: Point, extent: Extent) -> "Region":
        return Region(origin, Point(origin.x + extent.width, origin.y + extent.height))


# ---------------------------------------------------------------------------------------------
# C structs (layouts: include/dlimgedit/dlimgedit.h)

class _ImageView(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("channels", C.c_int), ("stride", C.c_int),
                ("pixels", C.c_void_p)]


class _Options(C.Structure):
    _fields_ = [("backend", C.c_int), ("model_directory", C.c_char_p)]


_u8pp = C.POINTER(C.c_void_p)

_API_FIELDS = [
    ("is_backend_supported", C.CFUNCTYPE(C.c_int, C.c_int)),
    ("create_environment", C.CFUNCTYPE(C.c_int, C.POINTER(C.c_void_p), C.POINTER(_Options))),
    ("destroy_environment", C.CFUNCTYPE(None, C.c_void_p)),
    ("process_image_for_segmentation", C.CFUNCTYPE(C.c_int, C.POINTER(C.c_void_p), C.POINTER(_ImageView), C.c_void_p)),
    ("get_segmentation_mask", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _u8pp,
                                          C.POINTER(C.c_float))),
    ("get_segmentation_extent", C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int))),
    ("destroy_segmentation", C.CFUNCTYPE(None, C.c_void_p)),
    ("segment_objects", C.CFUNCTYPE(C.c_int, C.POINTER(_ImageView), C.c_void_p, C.c_void_p)),
    ("load_image", C.CFUNCTYPE(C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _u8pp)),
    ("save_image", C.CFUNCTYPE(C.c_int, C.POINTER(_ImageView), C.c_char_p)),
    ("create_image", C.CFUNCTYPE(C.c_void_p, C.c_int, C.c_int, C.c_int)),
    ("destroy_image", C.CFUNCTYPE(None, C.c_void_p)),
    ("last_error", C.CFUNCTYPE(C.c_char_p)),
    # additions of this build
    ("process_images_for_segmentation", C.CFUNCTYPE(C.c_int, C.POINTER(C.c_void_p), C.POINTER(_ImageView), C.c_int,
                                                    C.c_void_p)),
    ("get_segmentation_masks", C.CFUNCTYPE(C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int),
                                           C.POINTER(C.c_int), _u8pp)),
]

REFERENCE_SLOTS = 13     # the reference's table ends after last_error


class _Api(C.Structure):
    _fields_ = _API_FIELDS


_lib = None
_api = None


def library() -> C.CDLL:
    """Loads libdlimgedit.so; raises if it has not been built (`python -m dlimgedit_amd.build`)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise Error(f"{LIB_PATH} not found: build it with `python -m dlimgedit_amd.build` "
                        "(the HIP library is the product; there is no Python fallback)")
        _lib = C.CDLL(str(LIB_PATH))
        _lib.dlimg_init.restype = C.POINTER(_Api)
        _lib.dlimg_init.argtypes = []
    return _lib


_hooks_lib = None


def hooks_library() -> C.CDLL:
    """Loads the library that exports the test / benchmark hooks (libdlimgedit_test.so; the tuning library when one is
    selected).  Raises if it has not been built: there is no fallback, a hook never runs anything but HIP code."""
    global _hooks_lib
    if _hooks_lib is None:
        if HOOKS_LIB_PATH == LIB_PATH:
            _hooks_lib = library()
        else:
            if not HOOKS_LIB_PATH.exists():
                raise Error(f"{HOOKS_LIB_PATH} not found: build it with `python -m dlimgedit_amd.build`")
            _hooks_lib = C.CDLL(str(HOOKS_LIB_PATH))
            _hooks_lib.dlimg_init.restype = C.POINTER(_Api)
            _hooks_lib.dlimg_init.argtypes = []
    return _hooks_lib


def _check_hook(result: int) -> None:
    """As _check, with the message of the library the hook lives in (last_error is per library and per thread)."""
    if result != 0:
        msg = hooks_library().dlimg_init().contents.last_error()
        raise Error(msg.decode("utf-8", "replace") if msg else "Unknown error")


def api() -> _Api:
    """dlimg::api(): the function table, initialised on first use (handle.hpp:27-34 in the reference)."""
    global _api
    if _api is None:
        _api = library().dlimg_init().contents
    return _api


def _check(result: int) -> None:
    if result != 0:
        msg = api().last_error()
        raise Error(msg.decode("utf-8", "replace") if msg else "Unknown error")


# ---------------------------------------------------------------------------------------------
# Image / ImageView

class ImageView:
    """Non-owning view of u8 pixels (numpy array kept alive by the view)."""

    def __init__(self, pixels: np.ndarray, channels: Channels = Channels.rgba, stride: Optional[int] = None):
        if pixels.dtype != np.uint8:
            raise TypeError("pixels must be uint8")
        if pixels.ndim == 2:
            pixels = pixels[:, :, None]
        h, w, c = pixels.shape
        if c != count(channels):
            raise ValueError(f"array has {c} channels, {channels!r} needs {count(channels)}")
        if stride is None:
            pixels = np.ascontiguousarray(pixels)
            stride = w * c
        self._array = pixels
        self.extent = Extent(w, h)
        self.channels = Channels(channels)
        self.stride = int(stride)

    def _c(self) -> _ImageView:
        return _ImageView(self.extent.width, self.extent.height, int(self.channels), self.stride,
                          self._array.ctypes.data)


class Image:
    """Owning image allocated by the library (create_image / destroy_image)."""

    def __init__(self, extent: Extent, channels: Channels = Channels.rgba):
        self._extent, self._channels = extent, Channels(channels)
        self._ptr = api().create_image(extent.width, extent.height, count(channels))
        if not self._ptr:
            raise Error("create_image failed")

    def extent(self) -> Extent:
        return self._extent

    def channels(self) -> Channels:
        return self._channels

    def size(self) -> int:
        return self._extent.width * self._extent.height * count(self._channels)

    def pixels(self) -> np.ndarray:
        buf = (C.c_uint8 * self.size()).from_address(self._ptr)
        buf._owner = self                        # the array keeps the Image (and with it the library's allocation) alive
        return np.frombuffer(buf, dtype=np.uint8).reshape(self._extent.height, self._extent.width, count(self._channels))

    def view(self) -> ImageView:
        return ImageView(self.pixels(), self._channels)

    @staticmethod
    def load(filepath) -> "Image":
        ext = (C.c_int * 2)()
        ch = C.c_int()
        px = C.c_void_p()
        _check(api().load_image(str(filepath).encode(), ext, C.byref(ch), C.byref(px)))
        img = Image.__new__(Image)               # adopt the library's allocation (freed by destroy_image)
        img._extent, img._channels, img._ptr = Extent(ext[0], ext[1]), Channels(ch.value), px.value
        return img

    @staticmethod
    def save(img: ImageView, filepath) -> None:
        v = img._c()
        _check(api().save_image(C.byref(v), str(filepath).encode()))

    def __del__(self):
        ptr, self._ptr = getattr(self, "_ptr", None), None
        if ptr and _api is not None:
            _api.destroy_image(ptr)


def _mask_image(extent: Extent) -> np.ndarray:
    """A result mask as the reference's wrapper makes it -- `Image(extent(), Channels::mask)`, i.e. memory from the library's
    create_image (dlimgedit.impl.hpp:84-88) -- seen as an (h, w) array that keeps the Image alive."""
    return Image(extent, Channels.mask).pixels().reshape(extent.height, extent.width)


# ---------------------------------------------------------------------------------------------
# Environment / Segmentation

@dataclass
class Options:
    backend: Backend = Backend.cpu
    model_directory: str = "models"


class Environment:
    """dlimg::Environment: owns the model cache; must outlive every Segmentation made from it."""

    @staticmethod
    def is_supported(backend: Backend) -> bool:
        return api().is_backend_supported(int(backend)) != 0

    def __init__(self, options: Options = Options()):
        self._handle = C.c_void_p()
        self._dir = str(options.model_directory).encode()
        opts = _Options(int(options.backend), self._dir)
        _check(api().create_environment(C.byref(self._handle), C.byref(opts)))

    def handle(self) -> C.c_void_p:
        return self._handle

    def close(self) -> None:
        h, self._handle = self._handle, C.c_void_p()
        if h and _api is not None:
            _api.destroy_environment(h)

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


@dataclass
class Mask:
    image: np.ndarray        # [H, W] uint8, values 0 or 255
    accuracy: float = 0.0


class Segmentation:
    """dlimg::Segmentation: cached image embedding + mask queries."""

    def __init__(self, handle: C.c_void_p, env: Environment):
        self._handle, self._env = handle, env   # keeps the environment alive

    @staticmethod
    def process(img: ImageView, env: Environment) -> "Segmentation":
        seg = Segmentation(C.c_void_p(), env)
        v = img._c()
        # the handle is assigned before encoding: on failure the wrapper still owns and frees it
        _check(api().process_image_for_segmentation(C.byref(seg._handle), C.byref(v), env.handle()))
        return seg

    @staticmethod
    def process_batch(imgs: Sequence[ImageView], env: Environment) -> list:
        n = len(imgs)
        handles = (C.c_void_p * n)()
        views = (_ImageView * n)(*[i._c() for i in imgs])
        segs = []
        try:
            _check(api().process_images_for_segmentation(handles, views, n, env.handle()))
        finally:
            segs = [Segmentation(C.c_void_p(h), env) for h in handles if h]
        return segs

    def extent(self) -> Extent:
        out = (C.c_int * 2)()
        api().get_segmentation_extent(self._handle, out)
        return Extent(out[0], out[1])

    def _query(self, point, region, n_masks, out=None):
        e = self.extent()
        masks = [_mask_image(e) for _ in range(n_masks)] if out is None else list(out)
        assert len(masks) == n_masks and all(m.shape == (e.height, e.width) and m.dtype == np.uint8 and
                                              m.flags.c_contiguous for m in masks)
        ptrs = (C.c_void_p * 3)(*([m.ctypes.data for m in masks] + [None] * (3 - n_masks)))
        acc = (C.c_float * 3)(0.0, 0.0, 0.0)
        p = (C.c_int * 2)(point.x, point.y) if point is not None else None
        r = (C.c_int * 4)(region.top_left.x, region.top_left.y, region.bottom_right.x, region.bottom_right.y) \
            if region is not None else None
        _check(api().get_segmentation_mask(self._handle, p, r, ptrs, acc))
        return masks, list(acc)

    def compute_mask(self, prompt, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Point -> best mask; Region -> mask of the largest object in the box.  `out`: the caller's own (h, w) uint8 buffer
        (the wrapper's compute_mask(point, uint8_t*) form) instead of a new Image of the library."""
        outs = None if out is None else [out]
        if isinstance(prompt, Point):
            return self._query(prompt, None, 1, outs)[0][0]
        if isinstance(prompt, Region):
            return self._query(None, prompt, 1, outs)[0][0]
        raise TypeError("prompt must be a Point or a Region")

    def compute_masks(self, point: Point) -> list:
        masks, acc = self._query(point, None, 3)
        return [Mask(m, a) for m, a in zip(masks, acc)]

    @staticmethod
    def compute_mask_batch(segs: Sequence["Segmentation"], points: Optional[Sequence[Point]] = None,
                           regions: Optional[Sequence[Region]] = None, out: Optional[Sequence[np.ndarray]] = None) -> list:
        n = len(segs)
        handles = (C.c_void_p * n)(*[s._handle for s in segs])
        outs = [_mask_image(s.extent()) for s in segs] if out is None else list(out)
        assert len(outs) == n and all(o.dtype == np.uint8 and o.flags.c_contiguous for o in outs)
        ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        p = r = None
        if points is not None:
            p = (C.c_int * (2 * n))(*[v for q in points for v in (q.x, q.y)])
        if regions is not None:
            r = (C.c_int * (4 * n))(*[v for q in regions for v in (q.top_left.x, q.top_left.y, q.bottom_right.x,
                                                                  q.bottom_right.y)])
        _check(api().get_segmentation_masks(handles, n, p, r, ptrs))
        return outs

    def close(self) -> None:
        h, self._handle = self._handle, C.c_void_p()
        if h and _api is not None:
            _api.destroy_segmentation(h)

    def __del__(self):
        self.close()


def segment_objects(img: ImageView, env: Environment) -> np.ndarray:
    out = np.empty((img.extent.height, img.extent.width), dtype=np.uint8)
    v = img._c()
    _check(api().segment_objects(C.byref(v), out.ctypes.data, env.handle()))
    return out


# ---------------------------------------------------------------------------------------------
# extension entry points (include/dlimgedit/dlimgedit_amd.h)

STAGES = ("pre", "gemm", "layernorm", "attention_window", "attention_global", "encoder_other", "decoder", "post",
          "gemm_stats", "gemm_norm", "gemm_norm_gelu", "gemm_other", "gemm_patch", "gemm_proj", "gemm_fc2")


class ext:
    _sigs_done = False
    _hook_sigs_done = False

    @staticmethod
    def _apply(lib, sig):
        for name, (args, res) in sig.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:           # an older library selected for an A/B (DLIMGEDIT_TUNING_LIB=<file>): the tools
                continue                     # that need the entry point fail when they call it; tests/test_abi.py checks them all
            fn.argtypes, fn.restype = args, res

    @classmethod
    def _l(cls):
        """The product library with the argument types of its extension entry points set."""
        lib = library()
        if not cls._sigs_done:
            vp, ci = C.c_void_p, C.c_int
            cls._apply(lib, {
                "dlimg_amd_device_count": ([], ci),
                "dlimg_amd_model_geometry": ([vp, C.POINTER(ci)], ci),
                "dlimg_amd_get_embedding": ([vp, vp], ci),
                "dlimg_amd_get_logits": ([vp, C.POINTER(ci), C.POINTER(ci), vp, vp], ci),
                "dlimg_amd_decoder_state": ([vp, C.POINTER(ci), vp, ci, C.c_char_p, ci], ci),
                "dlimg_amd_device_alloc": ([vp, C.c_size_t, C.POINTER(vp)], ci),
                "dlimg_amd_device_free": ([vp, vp], ci),
                "dlimg_amd_copy_to_device": ([vp, vp, vp, C.c_size_t], ci),
                "dlimg_amd_copy_to_host": ([vp, vp, vp, C.c_size_t], ci),
                "dlimg_amd_encode_and_mask": ([vp, C.POINTER(_ImageView), ci, C.POINTER(ci), C.POINTER(vp)], ci),
                "dlimg_amd_encode_only": ([vp, C.POINTER(_ImageView), ci], ci),
                "dlimg_amd_synchronize": ([vp], ci),
                "dlimg_amd_lane_count": ([vp], ci),
                "dlimg_amd_queue_config": ([vp, C.POINTER(ci)], ci),
                "dlimg_amd_image_memory": ([vp, C.c_size_t, C.POINTER(ci)], ci),
                "dlimg_amd_replica_count": ([vp], ci),
                "dlimg_amd_segmentation_device": ([vp, C.POINTER(ci), C.POINTER(ci)], ci),
                "dlimg_amd_get_segmentation_masks_device": ([C.POINTER(vp), ci, C.POINTER(ci), C.POINTER(ci), ci, vp,
                                                             C.POINTER(C.c_size_t)], ci),
                "dlimg_amd_set_profiling": ([vp, ci], ci),
                "dlimg_amd_take_stage_stats": ([vp, vp, vp, vp], ci),
                "dlimg_amd_birefnet_prepare_image": ([vp, ci, ci, ci, ci, vp, vp, vp], ci),
                "dlimg_amd_birefnet_process_mask": ([vp, ci, ci, vp], ci),
                "dlimg_amd_resize_mask": ([vp, ci, ci, ci, ci, ci, vp], ci),
            })
            cls._sigs_done = True
        return lib

    @classmethod
    def _h(cls):
        """The library with the test / benchmark hooks (hooks_library())."""
        lib = hooks_library()
        if not cls._hook_sigs_done:
            vp, ci, cf = C.c_void_p, C.c_int, C.c_float
            cls._apply(lib, {
                "dlimg_amd_test_mask_pieces": ([ci, C.POINTER(C.c_longlong), C.c_longlong, C.POINTER(C.c_longlong), ci,
                                               C.POINTER(C.c_longlong), ci, C.POINTER(ci)], ci),
                "dlimg_amd_test_plan_steps": ([ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), ci, ci, ci, ci, C.POINTER(ci), C.POINTER(ci), ci], ci),
                "dlimg_amd_test_parse_cpu_list": ([C.c_char_p, C.POINTER(ci), ci], ci),
                "dlimg_amd_test_preprocess": ([vp, ci, ci, ci, ci, vp], ci),
                "dlimg_amd_test_postprocess": ([vp, ci, vp, ci, ci, vp], ci),
                "dlimg_amd_test_postprocess_batch": ([vp, ci, ci, ci, vp], ci),
                "dlimg_amd_test_force_gemm_tile": ([ci], ci),
                "dlimg_amd_test_force_gemm_consumer_tile": ([ci], ci),
                "dlimg_amd_test_gemm": ([ci, ci, ci, vp, vp, vp, vp, ci, ci, vp, vp], ci),
                "dlimg_amd_test_gemm_ln": ([ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, cf, ci, vp, vp, vp], ci),
                "dlimg_amd_test_lane_worker": ([ci, ci, vp], ci),
                "dlimg_amd_test_gemm_stream": ([ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp], ci),
                "dlimg_amd_test_layernorm": ([vp, vp, vp, cf, ci, ci, ci, vp, vp], ci),
                "dlimg_amd_test_attention": ([ci, vp, vp, vp, vp, ci, ci, ci, vp], ci),
                "dlimg_amd_test_resize": ([vp, ci, ci, ci, ci, ci, ci, vp], ci),
                "dlimg_amd_bench_attention": ([ci, ci, ci, ci, ci, C.POINTER(C.c_double)], ci),
                "dlimg_amd_bench_prepost": ([ci, ci, ci, C.POINTER(C.c_double), C.POINTER(C.c_double)], ci),
                "dlimg_amd_bench_gemm": ([ci, ci, ci, ci, ci, ci, C.POINTER(C.c_double)], ci),
                "dlimg_amd_bench_gemm_streams": ([ci, ci, ci, ci, ci, ci, ci, ci, ci, C.POINTER(C.c_double)], ci),
                "dlimg_amd_bench_gemm_stamps": ([ci, ci, ci, ci, ci, ci, ci, ci, ci, C.POINTER(C.c_double), vp, ci], ci),
            })
            cls._hook_sigs_done = True
        return lib

    # entry points of the product library (include/dlimgedit/dlimgedit_amd.h; csrc/exports.map)
    EXPORTS = (
               "dlimg_amd_device_count", "dlimg_amd_model_geometry", "dlimg_amd_get_embedding",
               "dlimg_amd_get_logits", "dlimg_amd_decoder_state", "dlimg_amd_device_alloc", "dlimg_amd_device_free",
               "dlimg_amd_copy_to_device", "dlimg_amd_copy_to_host", "dlimg_amd_encode_and_mask",
               "dlimg_amd_encode_only", "dlimg_amd_synchronize", "dlimg_amd_lane_count", "dlimg_amd_queue_config",
               "dlimg_amd_image_memory",
               "dlimg_amd_replica_count", "dlimg_amd_segmentation_device", "dlimg_amd_get_segmentation_masks_device",
               "dlimg_amd_set_profiling", "dlimg_amd_take_stage_stats", "dlimg_amd_birefnet_prepare_image",
               "dlimg_amd_birefnet_process_mask", "dlimg_amd_resize_mask")
    # hooks of the test / tuning libraries (include/dlimgedit/dlimgedit_amd_test.h)
    HOOK_EXPORTS = (
               "dlimg_amd_test_mask_pieces", "dlimg_amd_test_plan_steps", "dlimg_amd_test_parse_cpu_list", "dlimg_amd_test_preprocess",
               "dlimg_amd_test_postprocess", "dlimg_amd_test_postprocess_batch", "dlimg_amd_test_force_gemm_tile",
               "dlimg_amd_test_force_gemm_consumer_tile", "dlimg_amd_test_gemm", "dlimg_amd_test_gemm_ln",
               "dlimg_amd_test_lane_worker", "dlimg_amd_test_gemm_stream", "dlimg_amd_test_layernorm",
               "dlimg_amd_test_attention", "dlimg_amd_test_resize", "dlimg_amd_bench_attention",
               "dlimg_amd_bench_prepost", "dlimg_amd_bench_gemm", "dlimg_amd_bench_gemm_streams",
               "dlimg_amd_bench_gemm_stamps")

    @staticmethod
    def _ptr(a: Optional[np.ndarray]):
        return None if a is None else a.ctypes.data

    @classmethod
    def device_count(cls) -> int:
        return cls._l().dlimg_amd_device_count()

    @classmethod
    def model_geometry(cls, env: Environment) -> Tuple[int, int, int, int]:
        out = (C.c_int * 4)()
        _check(cls._l().dlimg_amd_model_geometry(env.handle(), out))
        return tuple(out)

    @classmethod
    def get_embedding(cls, seg: Segmentation) -> np.ndarray:
        out = np.empty((4096, 256), dtype=np.float32)
        _check(cls._l().dlimg_amd_get_embedding(seg._handle, out.ctypes.data))
        return out

    @classmethod
    def get_logits(cls, seg: Segmentation, point: Optional[Point] = None, region: Optional[Region] = None):
        logits = np.empty((4, 256, 256), dtype=np.float32)
        iou = np.empty(4, dtype=np.float32)
        p = (C.c_int * 2)(point.x, point.y) if point is not None else None
        r = (C.c_int * 4)(region.top_left.x, region.top_left.y, region.bottom_right.x, region.bottom_right.y) \
            if region is not None else None
        _check(cls._l().dlimg_amd_get_logits(seg._handle, p, r, logits.ctypes.data, iou.ctypes.data))
        return logits, iou

    @classmethod
    def decoder_state(cls, seg: Segmentation, point: Point):
        """Diagnostic: {name: array} of the decoder's token-side workspaces after decoding `point`."""
        buf = C.create_string_buffer(1024)
        _check(cls._l().dlimg_amd_decoder_state(seg._handle, (C.c_int * 2)(point.x, point.y), None, 0, buf, 1024))
        parts = [(n, int(c)) for n, c in (item.split(":") for item in buf.value.decode().strip(",").split(","))]
        out = np.empty(sum(c for _, c in parts), dtype=np.float32)
        _check(cls._l().dlimg_amd_decoder_state(seg._handle, (C.c_int * 2)(point.x, point.y), out.ctypes.data, out.size, None, 0))
        res, off = {}, 0
        for n, c in parts:
            res[n] = out[off:off + c]
            off += c
        return res

    @classmethod
    def mask_pieces(cls, mask_bytes, extra_bytes: int = 0):
        """Host logic of the mask transfer (no GPU needed): (piece end offsets, [(piece, mask, staging offset, offset in the
        mask, bytes)])."""
        n = len(mask_bytes)
        sizes = (C.c_longlong * max(1, n))(*mask_bytes)
        ends = (C.c_longlong * 8)()
        cap = 8 * (n + 1)
        copies = (C.c_longlong * (5 * cap))()
        pieces = C.c_int(0)
        got = cls._h().dlimg_amd_test_mask_pieces(n, sizes, extra_bytes, ends, 8, copies, cap, C.byref(pieces))
        if got < 0:
            _check_hook(1)
        return [ends[i] for i in range(pieces.value)], [tuple(copies[5 * k + j] for j in range(5)) for k in range(got)]

    @classmethod
    def plan_steps(cls, passes_in_flight, images_in_flight, cursor: int, pending: int, width: int, depth: int, all_: bool):
        """Host logic of the device-step queue (no GPU needed): returns (passes [(lane, images)], new passes in flight, new
        images in flight, new cursor)."""
        lanes = len(passes_in_flight)
        p = (C.c_int * lanes)(*passes_in_flight)
        i = (C.c_int * lanes)(*images_in_flight)
        cur = C.c_int(cursor)
        cap = pending + lanes + 1
        out_lane, out_images = (C.c_int * cap)(), (C.c_int * cap)()
        n = cls._h().dlimg_amd_test_plan_steps(lanes, p, i, C.byref(cur), pending, width, depth, int(all_), out_lane, out_images, cap)
        if n < 0:
            _check_hook(1)
        return [(out_lane[k], out_images[k]) for k in range(n)], list(p), list(i), cur.value

    @classmethod
    def parse_cpu_list(cls, text: str) -> list:
        """Host logic (no GPU needed): sysfs cpulist text -> CPU indices, as the multi-GPU helper threads' binding reads it."""
        out = (C.c_int * 4096)()
        n = cls._h().dlimg_amd_test_parse_cpu_list(text.encode(), out, 4096)
        if n < 0:
            _check_hook(1)
        return list(out[:n])

    # -- benchmark path
    @classmethod
    def device_alloc(cls, env, nbytes: int) -> int:
        out = C.c_void_p()
        _check(cls._l().dlimg_amd_device_alloc(env.handle(), nbytes, C.byref(out)))
        return out.value

    @classmethod
    def device_free(cls, env, ptr: int) -> None:
        _check(cls._l().dlimg_amd_device_free(env.handle(), ptr))

    @classmethod
    def copy_to_device(cls, env, dst: int, src: np.ndarray) -> None:
        src = np.ascontiguousarray(src)
        _check(cls._l().dlimg_amd_copy_to_device(env.handle(), dst, src.ctypes.data, src.nbytes))

    @classmethod
    def copy_to_host(cls, env, dst: np.ndarray, src: int) -> None:
        _check(cls._l().dlimg_amd_copy_to_host(env.handle(), dst.ctypes.data, src, dst.nbytes))

    @staticmethod
    def device_views(ptrs, width, height, channels=Channels.rgba):
        n = len(ptrs)
        return (_ImageView * n)(*[_ImageView(width, height, int(channels), width * count(channels), p) for p in ptrs])

    @classmethod
    def encode_and_mask(cls, env, views, points, mask_ptrs) -> None:
        n = len(views)
        p = (C.c_int * (2 * n))(*[v for q in points for v in (q.x, q.y)])
        m = (C.c_void_p * n)(*mask_ptrs)
        _check(cls._l().dlimg_amd_encode_and_mask(env.handle(), views, n, p, m))

    @classmethod
    def encode_only(cls, env, views) -> None:
        _check(cls._l().dlimg_amd_encode_only(env.handle(), views, len(views)))

    @classmethod
    def synchronize(cls, env) -> None:
        _check(cls._l().dlimg_amd_synchronize(env.handle()))

    @classmethod
    def lane_count(cls, env) -> int:
        return cls._l().dlimg_amd_lane_count(env.handle())

    @classmethod
    def queue_config(cls, env) -> dict:
        """Effective settings of the step queue behind encode_and_mask, as the library clamped them."""
        out = (C.c_int * 6)()
        _check(cls._l().dlimg_amd_queue_config(env.handle(), out))
        return {"coalesce": out[0], "step_depth": out[1], "lanes": out[2], "lanes_in_use": out[3],
                "one_image_passes": out[4], "one_image_passes_alone": out[5]}

    @classmethod
    def image_memory_is_pinned(cls, array: np.ndarray) -> bool:
        """True when the array's bytes lie in pinned image memory of the library (read / written in place by the GPU)."""
        out = C.c_int(0)
        _check(cls._l().dlimg_amd_image_memory(array.ctypes.data, array.nbytes, C.byref(out)))
        return bool(out.value)

    @classmethod
    def replica_count(cls, env) -> int:
        return cls._l().dlimg_amd_replica_count(env.handle())

    @classmethod
    def segmentation_device(cls, seg) -> Tuple[int, int]:
        """(replica index in the environment's device list, HIP device index) holding the embedding."""
        r, d = C.c_int(), C.c_int()
        _check(cls._l().dlimg_amd_segmentation_device(seg._handle, C.byref(r), C.byref(d)))
        return r.value, d.value

    @classmethod
    def compute_mask_batch_device(cls, segs, dev_out: int, points=None, regions=None, root_device: int = 0) -> list:
        """Device-output form of Segmentation.compute_mask_batch: masks land tightly packed at `dev_out` (a device pointer
        on HIP device `root_device`), wherever their embeddings live; returns the byte offset of every mask."""
        n = len(segs)
        handles = (C.c_void_p * n)(*[s._handle for s in segs])
        p = r = None
        if points is not None:
            p = (C.c_int * (2 * n))(*[v for q in points for v in (q.x, q.y)])
        if regions is not None:
            r = (C.c_int * (4 * n))(*[v for q in regions for v in (q.top_left.x, q.top_left.y, q.bottom_right.x,
                                                                  q.bottom_right.y)])
        offsets = (C.c_size_t * n)()
        _check(cls._l().dlimg_amd_get_segmentation_masks_device(handles, n, p, r, root_device, dev_out, offsets))
        return list(offsets)

    @classmethod
    def set_profiling(cls, env, on: bool) -> None:
        _check(cls._l().dlimg_amd_set_profiling(env.handle(), int(on)))

    @classmethod
    def take_stage_stats(cls, env) -> dict:
        n = len(STAGES)
        ms, work, launches = (C.c_double * n)(), (C.c_double * n)(), (C.c_long * n)()
        _check(cls._l().dlimg_amd_take_stage_stats(env.handle(), ms, work, launches))
        return {s: {"ms": ms[i], "work": work[i], "launches": launches[i]} for i, s in enumerate(STAGES)}

    # -- single-kernel hooks
    @classmethod
    def test_preprocess(cls, pixels: np.ndarray, channels: Channels, stride: Optional[int] = None) -> np.ndarray:
        h, w = pixels.shape[:2]
        if stride is None:
            pixels = np.ascontiguousarray(pixels)
            stride = w * count(channels)
        out = np.empty((4096, 768), dtype=np.float16)
        _check_hook(cls._h().dlimg_amd_test_preprocess(pixels.ctypes.data, w, h, stride, int(channels), out.ctypes.data))
        return out

    @classmethod
    def test_postprocess(cls, planes: np.ndarray, out_w: int, out_h: int, iou: Optional[np.ndarray] = None) -> np.ndarray:
        planes = np.ascontiguousarray(planes, dtype=np.float32).reshape(-1, 256, 256)
        out = np.empty((out_h, out_w), dtype=np.uint8)
        iou = None if iou is None else np.ascontiguousarray(iou, dtype=np.float32)
        _check_hook(cls._h().dlimg_amd_test_postprocess(planes.ctypes.data, planes.shape[0], cls._ptr(iou), out_w, out_h,
                                                   out.ctypes.data))
        return out

    @classmethod
    def test_postprocess_batch(cls, planes: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
        """One launch for all planes (one mask each): the batched form of the post-processing kernel."""
        planes = np.ascontiguousarray(planes, dtype=np.float32).reshape(-1, 256, 256)
        out = np.empty((planes.shape[0], out_h, out_w), dtype=np.uint8)
        _check_hook(cls._h().dlimg_amd_test_postprocess_batch(planes.ctypes.data, planes.shape[0], out_w, out_h, out.ctypes.data))
        return out

    @classmethod
    def force_gemm_tile(cls, tile: int = -1, consumer_tile: int = -1) -> None:
        """Tile configuration the GEMM test hooks use wherever it fits (-1: the product's own choice); consumer_tile: a
        separate one for the LayerNorm-folded consumer of test_gemm_ln."""
        _check_hook(cls._h().dlimg_amd_test_force_gemm_tile(int(tile)))
        _check_hook(cls._h().dlimg_amd_test_force_gemm_consumer_tile(int(consumer_tile)))

    @classmethod
    def test_gemm(cls, A: np.ndarray, W: np.ndarray, bias=None, resid=None, act: int = 0, want_f16: bool = False):
        A = np.ascontiguousarray(A, dtype=np.float16)
        W = np.ascontiguousarray(W, dtype=np.float16)
        M, K = A.shape
        N = W.shape[0]
        bias = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        resid = None if resid is None else np.ascontiguousarray(resid, dtype=np.float32)
        out32 = np.empty((M, N), dtype=np.float32)
        out16 = np.empty((M, N), dtype=np.float16) if want_f16 else None
        _check_hook(cls._h().dlimg_amd_test_gemm(M, N, K, A.ctypes.data, W.ctypes.data, cls._ptr(bias), cls._ptr(resid),
                                            0 if resid is None else resid.shape[0], act, out32.ctypes.data,
                                            cls._ptr(out16)))
        return (out32, out16) if want_f16 else out32

    @classmethod
    def test_gemm_ln(cls, A1, W1, bias1, resid, W2, gamma, beta, bias2, eps: float, act: int = 0):
        """x = A1.W1^T + bias1 + resid; y = act(LayerNorm(x; gamma, beta).W2^T + bias2), with the LayerNorm folded into
        the second GEMM the way SamWeights does it (csrc/sam_model.cpp, Loader::linear_ln_h).  Returns (x, x as f16, y)."""
        A1 = np.ascontiguousarray(A1, dtype=np.float16)
        W1 = np.ascontiguousarray(W1, dtype=np.float16)
        M, K1 = A1.shape
        D = W1.shape[0]
        W2 = np.asarray(W2, dtype=np.float32)
        N = W2.shape[0]
        wg = np.ascontiguousarray((W2 * np.asarray(gamma, np.float32)[None, :]).astype(np.float16))
        colsum = np.ascontiguousarray(wg.astype(np.float64).sum(axis=1).astype(np.float32))
        b2 = np.ascontiguousarray((np.asarray(bias2, np.float64) + W2.astype(np.float64) @ np.asarray(beta, np.float64))
                                  .astype(np.float32))
        bias1 = None if bias1 is None else np.ascontiguousarray(bias1, dtype=np.float32)
        resid = None if resid is None else np.ascontiguousarray(resid, dtype=np.float32)
        x = np.empty((M, D), dtype=np.float32)
        xh = np.empty((M, D), dtype=np.float16)
        y = np.empty((M, N), dtype=np.float32)
        _check_hook(cls._h().dlimg_amd_test_gemm_ln(M, D, K1, N, A1.ctypes.data, W1.ctypes.data, cls._ptr(bias1),
                                               cls._ptr(resid), wg.ctypes.data, colsum.ctypes.data, b2.ctypes.data,
                                               eps, act, x.ctypes.data, xh.ctypes.data, y.ctypes.data))
        return x, xh, y

    @classmethod
    def test_lane_worker(cls, tasks: int, sleep_us: int = 0):
        """Runs the LaneWorker host-logic check; returns the order the tasks ran in (tasks + 1 entries when all ran)."""
        order = np.full(tasks + 1, -1, dtype=np.int32)
        ran = cls._h().dlimg_amd_test_lane_worker(tasks, sleep_us, order.ctypes.data)
        if ran < 0:
            _check_hook(1)
        return order[:ran].tolist()

    @classmethod
    def test_gemm_stream(cls, A, W, bias, resid_hi, resid_lo, pair: bool):
        """One stream-writing GEMM with row statistics; returns (x fp32 or None, hi, lo or None, stats[M, 48])."""
        A = np.ascontiguousarray(A, dtype=np.float16)
        W = np.ascontiguousarray(W, dtype=np.float16)
        M, K = A.shape
        D = W.shape[0]
        bias = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        resid_hi = None if resid_hi is None else np.ascontiguousarray(resid_hi, dtype=np.float16)
        resid_lo = None if resid_lo is None else np.ascontiguousarray(resid_lo, dtype=np.float16)
        x = None if pair else np.empty((M, D), dtype=np.float32)
        hi = np.empty((M, D), dtype=np.float16)
        lo = np.empty((M, D), dtype=np.float16) if pair else None
        stats = np.empty((M * 48,), dtype=np.float32)
        _check_hook(cls._h().dlimg_amd_test_gemm_stream(M, D, K, A.ctypes.data, W.ctypes.data, cls._ptr(bias), cls._ptr(resid_hi),
                                                   cls._ptr(resid_lo), int(pair), cls._ptr(x), hi.ctypes.data,
                                                   cls._ptr(lo), stats.ctypes.data))
        return x, hi, lo, stats

    @classmethod
    def test_layernorm(cls, x, w, b, eps: float, act: int = 0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        w = np.ascontiguousarray(w, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        rows, dim = x.shape
        o32 = np.empty_like(x)
        o16 = np.empty(x.shape, dtype=np.float16)
        _check_hook(cls._h().dlimg_amd_test_layernorm(x.ctypes.data, w.ctypes.data, b.ctypes.data, eps, rows, dim, act,
                                                 o32.ctypes.data, o16.ctypes.data))
        return o32, o16

    @classmethod
    def test_attention(cls, is_global: bool, qkv, qkv_bias, rel_h, rel_w, batch: int, heads: int, hd: int):
        qkv = np.ascontiguousarray(qkv, dtype=np.float16)
        rel_h = np.ascontiguousarray(rel_h, dtype=np.float32)
        rel_w = np.ascontiguousarray(rel_w, dtype=np.float32)
        qkv_bias = None if qkv_bias is None else np.ascontiguousarray(qkv_bias, dtype=np.float32)
        out = np.empty((batch * 4096, heads * hd), dtype=np.float16)
        _check_hook(cls._h().dlimg_amd_test_attention(int(is_global), qkv.ctypes.data, cls._ptr(qkv_bias), rel_h.ctypes.data,
                                                 rel_w.ctypes.data, batch, heads, hd, out.ctypes.data))
        return out

    @classmethod
    def test_resize(cls, pixels: np.ndarray, channels: Channels, out_w: int, out_h: int) -> np.ndarray:
        pixels = np.ascontiguousarray(pixels)
        h, w = pixels.shape[:2]
        c = count(channels)
        out = np.empty((out_h, out_w, c), dtype=np.uint8)
        _check_hook(cls._h().dlimg_amd_test_resize(pixels.ctypes.data, w, h, w * c, int(channels), out_w, out_h,
                                              out.ctypes.data))
        return out

    @classmethod
    def birefnet_prepare_image(cls, image: np.ndarray, channels: "Channels", mean, std) -> np.ndarray:
        """BiRefNet::prepare_image: u8 [H,W,C] -> f32 [1,3,H,W] (reference: segmentation.cpp:244-256)."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = image.shape[:2]
        mean = np.ascontiguousarray(mean, dtype=np.float32)
        std = np.ascontiguousarray(std, dtype=np.float32)
        out = np.empty((1, 3, h, w), dtype=np.float32)
        _check(cls._l().dlimg_amd_birefnet_prepare_image(image.ctypes.data, w, h, image.strides[0], int(channels),
                                                         mean.ctypes.data, std.ctypes.data, out.ctypes.data))
        return out

    @classmethod
    def birefnet_process_mask(cls, logits: np.ndarray) -> np.ndarray:
        """BiRefNet::process_mask: f32 [H,W] -> u8 [H,W] (reference: segmentation.cpp:258-270)."""
        logits = np.ascontiguousarray(logits, dtype=np.float32)
        h, w = logits.shape
        out = np.empty((h, w), dtype=np.uint8)
        _check(cls._l().dlimg_amd_birefnet_process_mask(logits.ctypes.data, w, h, out.ctypes.data))
        return out

    @classmethod
    def resize_mask(cls, mask: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
        """dlimg::resize_mask: u8 [H,W] -> u8 [out_h,out_w], box filter (reference: image.cpp:53-62)."""
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h, w = mask.shape
        out = np.empty((out_h, out_w), dtype=np.uint8)
        _check(cls._l().dlimg_amd_resize_mask(mask.ctypes.data, w, h, mask.strides[0], out_w, out_h, out.ctypes.data))
        return out

    @classmethod
    def bench_prepost(cls, batch: int = 16, iters: int = 50, working_set_mb: int = 768) -> Tuple[float, float]:
        """(ms per launch of the pre-processing kernel, of the post-processing kernel) on `batch` images / masks; launches
        rotate over `working_set_mb` MB of distinct inputs and outputs (nothing Infinity-Cache resident)."""
        pre, post = C.c_double(), C.c_double()
        _check_hook(cls._h().dlimg_amd_bench_prepost(batch, iters, working_set_mb, C.byref(pre), C.byref(post)))
        return pre.value, post.value

    @classmethod
    def bench_attention(cls, is_global: bool, heads: int = 12, hd: int = 64, batch: int = 1, iters: int = 50) -> float:
        """ms per launch of the encoder attention kernel alone, device-resident random data."""
        ms = C.c_double()
        _check_hook(cls._h().dlimg_amd_bench_attention(int(is_global), batch, heads, hd, iters, C.byref(ms)))
        return ms.value

    @classmethod
    def bench_gemm(cls, M: int, N: int, K: int, act: int = 0, iters: int = 20, flavour: int = 0, tile: int = -1,
                   shared: bool = False, streams: int = 1) -> float:
        ms = C.c_double()
        _check_hook(cls._h().dlimg_amd_bench_gemm_streams(M, N, K, act, flavour, tile, int(shared), streams, iters,
                                                     C.byref(ms)))
        return ms.value

    @classmethod
    def bench_gemm_stamps(cls, M: int, N: int, K: int, act: int = 0, iters: int = 20, flavour: int = 0, tile: int = 9,
                          streams: int = 1, groups: int = 4096):
        """(ms per GEMM, stamps [groups][4] u64: main-loop cycles, main-loop 100 MHz ticks, kernel cycles, kernel ticks)."""
        ms = C.c_double()
        st = np.zeros((groups, 4), np.uint64)
        _check_hook(cls._h().dlimg_amd_bench_gemm_stamps(M, N, K, act, flavour, tile, 0, streams, iters, C.byref(ms),
                                                    st.ctypes.data, groups))
        return ms.value, st
