"""ctypes view of the two product libraries (test / bench plumbing, not the product).

The product is native: ``csrc/librt_hip.so`` (HIP kernels + the C ABI of ``include/rt_api.h``)
and ``csrc/librt_host.so`` (scene ingest, ``include/rt_host.h``).  This module only loads them and
declares their prototypes so Python tests and ``bench.py`` can call through the C ABI.  There is
no Python or CPU fallback for the render path: if ``librt_hip.so`` is missing, ``load_hip()``
raises.

The directory name contains a hyphen, so import it through ``__graft_entry__.load_package()``
(which registers it as ``opengl_raytracer_amd``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
REPO = os.path.dirname(_HERE)

RT_FLAG_FIX_INVDIR = 0x1
RT_FLAG_SUM = 0x2
RT_FLAG_MEGAKERNEL = 0x4
RT_FLAG_COUNTERS = 0x8
RT_FLAG_WAVEFRONT = 0x10
RT_FLAG_REUSE_PRIMARY = 0x20
RT_FLAG_NO_CULL = 0x40
RT_FLAG_PACKETS = 0x80
RT_BAND_INTERLEAVED = 0x8000
RT_NUM_COUNTERS = 6
RT_NUM_KERNEL_CLASSES = 8
KERNEL_CLASS_NAMES = ["generate", "extend", "shade", "tail", "accumulate", "mega", "paths", "primary"]

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint32_p = C.POINTER(C.c_uint32)


class HostScene(C.Structure):
    """rt_host_scene (include/rt_host.h)."""

    _fields_ = [
        ("num_tris", C.c_int32),
        ("num_nodes", C.c_int32),
        ("num_vertices", C.c_int32),
        ("num_leaves", C.c_int32),
        ("min_leaf_size", C.c_int32),
        ("max_leaf_size", C.c_int32),
        ("max_depth", C.c_int32),
        ("reserved", C.c_int32),
        ("nodes", c_float_p),
        ("triv0", c_float_p),
        ("triv1", c_float_p),
        ("triv2", c_float_p),
        ("tri_src", c_int32_p),
        ("links", c_int32_p),
    ]


class RtScene(C.Structure):
    """rt_scene (include/rt_api.h)."""

    _fields_ = [
        ("num_nodes", C.c_int32),
        ("num_tris", C.c_int32),
        ("nodes", c_float_p),
        ("triv0", c_float_p),
        ("triv1", c_float_p),
        ("triv2", c_float_p),
        ("colours", c_float_p),
        ("links", c_int32_p),
    ]


class RtCamera(C.Structure):
    _fields_ = [
        ("pos", C.c_float * 3),
        ("rot", C.c_float * 9),
        ("fov_rad", C.c_float),
        ("viewport_dist", C.c_float),
    ]


class RtParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("rays_per_pixel", C.c_uint32),
        ("ray_bounces", C.c_uint32),
        ("render_mode", C.c_uint32),
        ("frame_base", C.c_uint32),
        ("frames", C.c_uint32),
        ("frame_stride", C.c_uint32),
        ("flags", C.c_uint32),
        ("band", C.c_uint32),
    ]


# every symbol include/rt_host.h declares
HOST_SYMBOLS = [
    "rt_host_read_obj",
    "rt_host_free",
    "rt_host_build_scene",
    "rt_host_load_scene",
    "rt_host_free_scene",
    "rt_host_make_colours",
    "rt_host_camera_rotation",
    "rt_host_hash64",
]

# every symbol include/rt_api.h declares
HIP_SYMBOLS = [
    "rt_create",
    "rt_destroy",
    "rt_last_error",
    "rt_set_stream",
    "rt_get_stream",
    "rt_set_frames_ahead",
    "rt_synchronize",
    "rt_upload_scene",
    "rt_get_normals",
    "rt_build_scene_device",
    "rt_rebuild_scene_device",
    "rt_refit_scene_device",
    "rt_get_built_scene",
    "rt_get_built_links",
    "rt_raytrace_frame",
    "rt_raytrace_frame_noalloc",
    "rt_render_device",
    "rt_render",
    "rt_primary_hits",
    "rt_counters",
    "rt_reset_counters",
    "rt_set_profiling",
    "rt_kernel_times",
    "rt_selftest_math",
    "rt_version",
]

_host = None
_hip = None
_hip_tune = None


def load_host() -> C.CDLL:
    global _host
    if _host is None:
        path = os.path.join(CSRC, "librt_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run __graft_entry__.build() (make -C {CSRC})")
        lib = C.CDLL(path)
        lib.rt_host_read_obj.argtypes = [C.c_char_p, C.POINTER(c_float_p), c_int32_p, C.POINTER(c_uint32_p), c_int32_p]
        lib.rt_host_read_obj.restype = C.c_int
        lib.rt_host_free.argtypes = [C.c_void_p]
        lib.rt_host_free.restype = None
        lib.rt_host_build_scene.argtypes = [c_float_p, C.c_int32, c_uint32_p, C.c_int32, C.POINTER(HostScene)]
        lib.rt_host_build_scene.restype = C.c_int
        lib.rt_host_load_scene.argtypes = [C.c_char_p, C.POINTER(HostScene)]
        lib.rt_host_load_scene.restype = C.c_int
        lib.rt_host_free_scene.argtypes = [C.POINTER(HostScene)]
        lib.rt_host_free_scene.restype = None
        lib.rt_host_make_colours.argtypes = [C.c_uint32, C.c_int32, c_float_p]
        lib.rt_host_make_colours.restype = None
        lib.rt_host_camera_rotation.argtypes = [C.c_double, C.c_double, c_float_p]
        lib.rt_host_camera_rotation.restype = None
        lib.rt_host_hash64.argtypes = [C.c_void_p, C.c_uint64]
        lib.rt_host_hash64.restype = C.c_uint64
        _host = lib
    return _host


def load_hip(tunable: bool = False) -> C.CDLL:
    """Loads librt_hip.so; raises if it has not been built (no fallback exists).  tunable=True loads librt_hip_tune.so
    instead: the same sources built with -DRT_TUNABLES, the only build that reads RT_TUNE_* / RT_BUILD_* from the
    environment (tools and the tests that cross-check a non-default variant; both libraries can be loaded side by side)."""
    global _hip, _hip_tune
    if (_hip_tune if tunable else _hip) is None:
        # RT_HIP_LIB lets a developer load an instrumented build of the same library
        path = os.environ.get("RT_HIP_LIB") or os.path.join(CSRC, "librt_hip_tune.so" if tunable else "librt_hip.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run __graft_entry__.build() (make -C {CSRC})")
        lib = C.CDLL(path)
        vp = C.c_void_p
        lib.rt_create.argtypes = [C.c_int, C.POINTER(vp)]
        lib.rt_destroy.argtypes = [vp]
        lib.rt_last_error.argtypes = [vp]
        lib.rt_last_error.restype = C.c_char_p
        lib.rt_set_stream.argtypes = [vp, vp]
        lib.rt_get_stream.argtypes = [vp, C.POINTER(vp)]
        lib.rt_set_frames_ahead.argtypes = [vp, C.c_int32]
        lib.rt_synchronize.argtypes = [vp]
        lib.rt_upload_scene.argtypes = [vp, C.POINTER(RtScene)]
        lib.rt_get_normals.argtypes = [vp, c_float_p]
        lib.rt_build_scene_device.argtypes = [vp, c_float_p, C.c_int32, c_uint32_p, C.c_int32, c_float_p]
        lib.rt_get_built_links.argtypes = [vp, c_int32_p]
        lib.rt_rebuild_scene_device.argtypes = [vp, vp]
        lib.rt_refit_scene_device.argtypes = [vp, vp]
        lib.rt_get_built_scene.argtypes = [vp, c_int32_p, c_int32_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                           c_int32_p, c_float_p]
        lib.rt_raytrace_frame.argtypes = [vp, C.POINTER(RtCamera), C.POINTER(RtParams), C.c_uint32, vp, vp]
        lib.rt_raytrace_frame_noalloc.argtypes = [vp, C.POINTER(RtCamera), C.POINTER(RtParams), C.c_uint32, vp, vp]
        lib.rt_render_device.argtypes = [vp, C.POINTER(RtCamera), C.POINTER(RtParams), vp]
        lib.rt_render.argtypes = [vp, C.POINTER(RtCamera), C.POINTER(RtParams), c_float_p]
        lib.rt_primary_hits.argtypes = [vp, C.POINTER(RtCamera), C.c_int32, C.c_int32, c_int32_p, c_float_p]
        lib.rt_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
        lib.rt_reset_counters.argtypes = [vp]
        lib.rt_set_profiling.argtypes = [vp, C.c_int]
        lib.rt_kernel_times.argtypes = [vp, c_float_p, c_uint32_p]
        lib.rt_selftest_math.argtypes = [vp, c_float_p, C.c_int32, c_float_p]
        lib.rt_version.argtypes = []
        lib.rt_version.restype = C.c_char_p
        for name in HIP_SYMBOLS:
            fn = getattr(lib, name)
            if name not in ("rt_last_error", "rt_version"):
                fn.restype = C.c_int
        if tunable:
            _hip_tune = lib
        else:
            _hip = lib
    return _hip_tune if tunable else _hip


def build_id() -> str:
    """The source hash librt_hip.so was built from (second-to-last field group of rt_version())."""
    parts = load_hip().rt_version().decode().split()
    return parts[2] if len(parts) >= 3 else "unknown"


class DeviceBuffer:
    """Plain device memory from the HIP runtime the render library is linked against, for callers that hold no other
    GPU library (the tests).  Allocated on the current HIP device."""

    _rt = None

    @classmethod
    def _runtime(cls):
        if cls._rt is None:
            load_hip()                                   # makes sure the runtime below is the one librt_hip.so uses
            rt = C.CDLL("libamdhip64.so.7")
            rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            rt.hipFree.argtypes = [C.c_void_p]
            rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            cls._rt = rt
        return cls._rt

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        if self._runtime().hipMalloc(C.byref(p), self.nbytes) != 0 or not p.value:
            raise RuntimeError(f"hipMalloc of {nbytes} bytes failed")
        self.ptr = p.value

    @classmethod
    def from_array(cls, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes)
        if cls._runtime().hipMemcpy(buf.ptr, a.ctypes.data, a.nbytes, 1) != 0:   # hipMemcpyHostToDevice
            raise RuntimeError("hipMemcpy to the device failed")
        return buf

    def to_array(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        if self._runtime().hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2) != 0:  # hipMemcpyDeviceToHost
            raise RuntimeError("hipMemcpy from the device failed")
        return out

    def free(self):
        if self.ptr:
            self._runtime().hipFree(self.ptr)
            self.ptr = None


def fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_float_p)


class Scene:
    """A scene in the reference's upload layout, as numpy arrays (host memory)."""

    def __init__(self, nodes, triv0, triv1, triv2, colours, tri_src=None, stats=None, links=None):
        self.nodes = np.ascontiguousarray(nodes, dtype=np.float32).reshape(-1, 12)
        self.triv0 = np.ascontiguousarray(triv0, dtype=np.float32).reshape(-1, 4)
        self.triv1 = np.ascontiguousarray(triv1, dtype=np.float32).reshape(-1, 4)
        self.triv2 = np.ascontiguousarray(triv2, dtype=np.float32).reshape(-1, 4)
        self.colours = np.ascontiguousarray(colours, dtype=np.float32).reshape(-1, 4)
        self.tri_src = tri_src
        self.stats = stats or {}
        # optional exact (a, b) per node, rt_scene.links: required beyond 2^24 nodes, where float ids are inexact
        self.links = None if links is None else np.ascontiguousarray(links, dtype=np.int32).reshape(-1, 2)

    @property
    def num_tris(self):
        return self.triv0.shape[0]

    @property
    def num_nodes(self):
        return self.nodes.shape[0]

    def as_rt_scene(self) -> RtScene:
        return RtScene(
            self.num_nodes, self.num_tris, fptr(self.nodes), fptr(self.triv0), fptr(self.triv1), fptr(self.triv2),
            fptr(self.colours), None if self.links is None else self.links.ctypes.data_as(c_int32_p)
        )


def make_colours(seed: int, num_tris: int) -> np.ndarray:
    out = np.empty((num_tris, 4), dtype=np.float32)
    load_host().rt_host_make_colours(seed, num_tris, fptr(out))
    return out


def _scene_from_host(hs: HostScene, colour_seed: int) -> Scene:
    n, t = hs.num_nodes, hs.num_tris
    nodes = np.ctypeslib.as_array(hs.nodes, shape=(n, 12)).copy()
    v0 = np.ctypeslib.as_array(hs.triv0, shape=(t, 4)).copy()
    v1 = np.ctypeslib.as_array(hs.triv1, shape=(t, 4)).copy()
    v2 = np.ctypeslib.as_array(hs.triv2, shape=(t, 4)).copy()
    src = np.ctypeslib.as_array(hs.tri_src, shape=(t,)).copy()
    links = np.ctypeslib.as_array(hs.links, shape=(n, 2)).copy()
    stats = {
        "num_leaves": hs.num_leaves,
        "min_leaf_size": hs.min_leaf_size,
        "max_leaf_size": hs.max_leaf_size,
        "max_depth": hs.max_depth,
        "num_vertices": hs.num_vertices,
    }
    return Scene(nodes, v0, v1, v2, make_colours(colour_seed, t), src, stats, links)


def load_scene(obj_path: str, colour_seed: int = 1) -> Scene:
    """loadScene() of the reference for an explicit path, through librt_host.so."""
    lib = load_host()
    hs = HostScene()
    rc = lib.rt_host_load_scene(obj_path.encode(), C.byref(hs))
    if rc != 0:
        raise RuntimeError(f"rt_host_load_scene({obj_path}) failed with {rc}")
    try:
        return _scene_from_host(hs, colour_seed)
    finally:
        lib.rt_host_free_scene(C.byref(hs))


def build_scene(vertices: np.ndarray, faces: np.ndarray, colour_seed: int = 1) -> Scene:
    lib = load_host()
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
    hs = HostScene()
    rc = lib.rt_host_build_scene(fptr(v), v.shape[0], f.ctypes.data_as(c_uint32_p), f.shape[0], C.byref(hs))
    if rc != 0:
        raise RuntimeError(f"rt_host_build_scene failed with {rc}")
    try:
        return _scene_from_host(hs, colour_seed)
    finally:
        lib.rt_host_free_scene(C.byref(hs))


def camera_rotation(pitch: float, yaw: float) -> np.ndarray:
    out = np.empty(9, dtype=np.float32)
    load_host().rt_host_camera_rotation(pitch, yaw, fptr(out))
    return out


def make_camera(pos=(0.0, 0.0, -2.0), pitch=0.0, yaw=0.0, fov_deg=90.0, viewport_dist=0.1) -> RtCamera:
    """Camera with the reference's defaults (constants.h:21-22,30; raytrace.cpp:69-70)."""
    cam = RtCamera()
    cam.pos[:] = [float(x) for x in pos]
    cam.rot[:] = [float(x) for x in camera_rotation(pitch, yaw)]
    cam.fov_rad = float(np.float32(float(fov_deg) * np.pi / 180.0))  # evaluated in double, then narrowed
    cam.viewport_dist = float(np.float32(viewport_dist))
    return cam


def make_params(width=1920, height=1080, rays_per_pixel=1, ray_bounces=100, render_mode=1, frame_base=0, frames=1,
                frame_stride=1, flags=0, band=None, interleaved=False) -> RtParams:
    """band = (index, count) restricts the render to one horizontal band of the image (tile partition); with
    interleaved=True the band is every count-th stripe of 16 pixel rows instead of one contiguous range."""
    b = 0 if band is None else ((int(band[0]) << 16) | int(band[1]) | (RT_BAND_INTERLEAVED if interleaved else 0))
    return RtParams(width, height, rays_per_pixel, ray_bounces, render_mode, frame_base, frames, frame_stride, flags, b)


class RenderContext:
    """Thin owner of an rt_context; every method goes straight through the C ABI."""

    def __init__(self, device_id: int = 0, tunable: bool = False):
        self.lib = load_hip(tunable)
        self.ctx = C.c_void_p()
        rc = self.lib.rt_create(device_id, C.byref(self.ctx))
        if rc != 0:
            raise RuntimeError(f"rt_create failed ({rc}): {self.lib.rt_last_error(None).decode()}")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.rt_last_error(self.ctx).decode()}")

    def close(self):
        if self.ctx:
            self.lib.rt_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle: int):
        self._check(self.lib.rt_set_stream(self.ctx, C.c_void_p(stream_handle)), "rt_set_stream")

    def get_stream(self) -> int:
        h = C.c_void_p()
        self._check(self.lib.rt_get_stream(self.ctx, C.byref(h)), "rt_get_stream")
        return h.value or 0

    def set_frames_ahead(self, max_frames: int):
        """Most frames raytrace_frame traces ahead of a steady frame-by-frame loop per launch (0: never)."""
        self._check(self.lib.rt_set_frames_ahead(self.ctx, max_frames), "rt_set_frames_ahead")

    def synchronize(self):
        self._check(self.lib.rt_synchronize(self.ctx), "rt_synchronize")

    def upload_scene(self, scene: Scene):
        s = scene.as_rt_scene()
        self._check(self.lib.rt_upload_scene(self.ctx, C.byref(s)), "rt_upload_scene")
        self.num_tris = scene.num_tris

    def build_scene_device(self, vertices: np.ndarray, faces: np.ndarray, face_colours: np.ndarray, download: bool = True):
        """GPU binned-SAH build; returns the built scene downloaded in the reference layout (a Scene, or None with
        download=False) and the device build time in milliseconds."""
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
        col = np.ascontiguousarray(face_colours, dtype=np.float32).reshape(-1, 4)
        assert col.shape[0] == f.shape[0]
        self._check(self.lib.rt_build_scene_device(self.ctx, fptr(v), v.shape[0], f.ctypes.data_as(c_uint32_p), f.shape[0],
                                                   fptr(col)), "rt_build_scene_device")
        self.num_tris = f.shape[0]
        return self.get_built_scene(download)

    def rebuild_scene_device(self, d_vertices_ptr: int, download: bool = False):
        """Rebuild with the faces and colours of the last build_scene_device and new vertex positions read from a
        device pointer (num_vertices x 3 floats)."""
        self._check(self.lib.rt_rebuild_scene_device(self.ctx, C.c_void_p(d_vertices_ptr)), "rt_rebuild_scene_device")
        return self.get_built_scene(download)

    def refit_scene_device(self, d_vertices_ptr: int, download: bool = False):
        """Keep the tree of the last build, recompute its boxes from new vertex positions (device pointer)."""
        self._check(self.lib.rt_refit_scene_device(self.ctx, C.c_void_p(d_vertices_ptr)), "rt_refit_scene_device")
        return self.get_built_scene(download)

    def get_built_scene(self, download: bool = True):
        n, t, ms = C.c_int32(), C.c_int32(), C.c_float()
        self._check(self.lib.rt_get_built_scene(self.ctx, C.byref(n), C.byref(t), None, None, None, None, None, None, C.byref(ms)),
                    "rt_get_built_scene")
        if not download:
            return None, ms.value
        nodes = np.empty((n.value, 12), dtype=np.float32)
        v0, v1, v2, cc = (np.empty((t.value, 4), dtype=np.float32) for _ in range(4))
        src = np.empty(t.value, dtype=np.int32)
        self._check(self.lib.rt_get_built_scene(self.ctx, None, None, fptr(nodes), fptr(v0), fptr(v1), fptr(v2), fptr(cc),
                                                src.ctypes.data_as(c_int32_p), None), "rt_get_built_scene")
        links = np.empty((n.value, 2), dtype=np.int32)
        self._check(self.lib.rt_get_built_links(self.ctx, links.ctypes.data_as(c_int32_p)), "rt_get_built_links")
        return Scene(nodes, v0, v1, v2, cc, src, {"build_ms": ms.value}, links), ms.value

    def get_normals(self) -> np.ndarray:
        out = np.empty((self.num_tris, 4), dtype=np.float32)
        self._check(self.lib.rt_get_normals(self.ctx, fptr(out)), "rt_get_normals")
        return out

    def render(self, cam: RtCamera, params: RtParams, accum: np.ndarray | None = None) -> np.ndarray:
        if accum is None:
            accum = np.zeros((params.height, params.width, 4), dtype=np.float32)
        self._check(self.lib.rt_render(self.ctx, C.byref(cam), C.byref(params), fptr(accum)), "rt_render")
        return accum

    def render_device(self, cam: RtCamera, params: RtParams, d_accum_ptr: int):
        self._check(
            self.lib.rt_render_device(self.ctx, C.byref(cam), C.byref(params), C.c_void_p(d_accum_ptr)),
            "rt_render_device",
        )

    def raytrace_frame(self, cam, params, frame_count: int, d_prev_ptr: int, d_out_ptr: int):
        self._check(
            self.lib.rt_raytrace_frame(
                self.ctx, C.byref(cam), C.byref(params), frame_count, C.c_void_p(d_prev_ptr), C.c_void_p(d_out_ptr)
            ),
            "rt_raytrace_frame",
        )

    def raytrace_frame_noalloc(self, cam, params, frame_count: int, d_prev_ptr: int, d_out_ptr: int):
        self._check(
            self.lib.rt_raytrace_frame_noalloc(
                self.ctx, C.byref(cam), C.byref(params), frame_count, C.c_void_p(d_prev_ptr), C.c_void_p(d_out_ptr)
            ),
            "rt_raytrace_frame_noalloc",
        )

    def primary_hits(self, cam: RtCamera, width: int, height: int):
        ids = np.empty((height, width), dtype=np.int32)
        t = np.empty((height, width), dtype=np.float32)
        self._check(
            self.lib.rt_primary_hits(self.ctx, C.byref(cam), width, height, ids.ctypes.data_as(c_int32_p), fptr(t)),
            "rt_primary_hits",
        )
        return ids, t

    def counters(self) -> np.ndarray:
        out = (C.c_uint64 * RT_NUM_COUNTERS)()
        self._check(self.lib.rt_counters(self.ctx, out), "rt_counters")
        return np.array(list(out), dtype=np.uint64)

    def reset_counters(self):
        self._check(self.lib.rt_reset_counters(self.ctx), "rt_reset_counters")

    def set_profiling(self, enable: bool):
        self._check(self.lib.rt_set_profiling(self.ctx, int(enable)), "rt_set_profiling")

    def kernel_times(self):
        ms = np.zeros(RT_NUM_KERNEL_CLASSES, dtype=np.float32)
        n = np.zeros(RT_NUM_KERNEL_CLASSES, dtype=np.uint32)
        self._check(self.lib.rt_kernel_times(self.ctx, fptr(ms), n.ctypes.data_as(c_uint32_p)), "rt_kernel_times")
        return ms, n

    def selftest_math(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty((6, x.size), dtype=np.float32)
        self._check(self.lib.rt_selftest_math(self.ctx, fptr(x), x.size, fptr(out)), "rt_selftest_math")
        return out
