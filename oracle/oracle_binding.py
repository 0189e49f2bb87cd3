"""ctypes view of the CPU oracle (oracle/librt_oracle.so).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the
product.  The oracle restates the reference's render path (see the header of rt_oracle.cpp).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_variant = None

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint32_p = C.POINTER(C.c_uint32)
c_u64_p = C.POINTER(C.c_uint64)


def load(variant: bool = False) -> C.CDLL:
    """librt_oracle.so, or with variant=True librt_oracle_variant.so: the same code over another admissible reading of the
    GLSL built-ins (rt_math_variant.h), for measuring how far the unpinned part of the numerics contract can move an image."""
    global _lib, _variant
    if (_variant if variant else _lib) is None:
        path = os.path.join(_HERE, "librt_oracle_variant.so" if variant else "librt_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: make -C {_HERE}")
        lib = C.CDLL(path)
        vp = C.c_void_p
        lib.oracle_normals.argtypes = [vp, c_float_p]
        lib.oracle_normals.restype = None
        lib.oracle_frame_rows.argtypes = [vp, c_float_p, vp, vp, C.c_uint32, c_float_p, c_float_p, c_u64_p, C.c_int,
                                          C.c_int, C.c_int]
        lib.oracle_frame_rows.restype = None
        lib.oracle_render.argtypes = [vp, vp, vp, c_float_p, c_u64_p, C.c_int]
        lib.oracle_render.restype = None
        lib.oracle_primary_hits.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, c_int32_p, c_float_p, C.c_int]
        lib.oracle_primary_hits.restype = None
        lib.oracle_rand.argtypes = [C.c_uint32, C.c_int, c_uint32_p, c_float_p]
        lib.oracle_rand.restype = None
        lib.oracle_tri_distance.argtypes = [c_float_p] * 5
        lib.oracle_tri_distance.restype = C.c_float
        lib.oracle_box_distance.argtypes = [c_float_p] * 4
        lib.oracle_box_distance.restype = C.c_float
        lib.oracle_sky.argtypes = [c_float_p, c_float_p]
        lib.oracle_sky.restype = None
        lib.oracle_hemisphere_dir.argtypes = [c_uint32_p, c_float_p, c_float_p]
        lib.oracle_hemisphere_dir.restype = None
        lib.oracle_primary_ray.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p]
        lib.oracle_primary_ray.restype = None
        lib.oracle_math.argtypes = [c_float_p, C.c_int, c_float_p]
        lib.oracle_math.restype = None
        lib.oracle_max_threads.argtypes = []
        lib.oracle_max_threads.restype = C.c_int
        if variant:
            _variant = lib
        else:
            _lib = lib
    return _variant if variant else _lib


def _f(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_float_p)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def normals(scene) -> np.ndarray:
    s = scene.as_rt_scene()
    out = np.empty((scene.num_tris, 4), dtype=np.float32)
    load().oracle_normals(C.byref(s), _f(out))
    return out


def render(scene, cam, params, accum=None, threads=0, variant=False):
    """Returns (rgba[H,W,4], counters[6])."""
    s = scene.as_rt_scene()
    if accum is None:
        accum = np.zeros((params.height, params.width, 4), dtype=np.float32)
    counters = (C.c_uint64 * 6)()
    load(variant).oracle_render(C.byref(s), C.byref(cam), C.byref(params), _f(accum), counters, threads)
    return accum, np.array(list(counters), dtype=np.uint64)


def frame_rows(scene, nrm, cam, params, frame_count, out, row_begin, row_end, threads=0):
    s = scene.as_rt_scene()
    counters = (C.c_uint64 * 6)()
    load().oracle_frame_rows(C.byref(s), _f(nrm), C.byref(cam), C.byref(params), frame_count, _f(out), _f(out),
                             counters, row_begin, row_end, threads)
    return np.array(list(counters), dtype=np.uint64)


def primary_hits(scene, cam, width, height, brute_force=False, threads=0, variant=False):
    s = scene.as_rt_scene()
    ids = np.empty((height, width), dtype=np.int32)
    t = np.empty((height, width), dtype=np.float32)
    load(variant).oracle_primary_hits(C.byref(s), C.byref(cam), width, height, int(brute_force),
                               ids.ctypes.data_as(c_int32_p), _f(t), threads)
    return ids, t


def rand(seed: int, n: int):
    raw = np.empty(n, dtype=np.uint32)
    flt = np.empty(n, dtype=np.float32)
    load().oracle_rand(seed, n, raw.ctypes.data_as(c_uint32_p), _f(flt))
    return raw, flt


def tri_distance(origin, direction, a, b, c) -> float:
    return load().oracle_tri_distance(_f3(origin), _f3(direction), _f3(a), _f3(b), _f3(c))


def box_distance(origin, inv_dir, bmin, bmax) -> float:
    return load().oracle_box_distance(_f3(origin), _f3(inv_dir), _f3(bmin), _f3(bmax))


def sky(direction) -> np.ndarray:
    out = (C.c_float * 3)()
    load().oracle_sky(_f3(direction), out)
    return np.array(list(out), dtype=np.float32)


def hemisphere_dir(seed: int, normal):
    s = C.c_uint32(seed)
    out = (C.c_float * 3)()
    load().oracle_hemisphere_dir(C.byref(s), _f3(normal), out)
    return np.array(list(out), dtype=np.float32), s.value


def primary_ray(cam, width, height, gx, gy):
    d = (C.c_float * 3)()
    inv = (C.c_float * 3)()
    load().oracle_primary_ray(C.byref(cam), width, height, gx, gy, d, inv)
    return np.array(list(d), dtype=np.float32), np.array(list(inv), dtype=np.float32)


def math(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty((6, x.size), dtype=np.float32)
    load().oracle_math(_f(x), x.size, _f(out))
    return out


def max_threads() -> int:
    return load().oracle_max_threads()
