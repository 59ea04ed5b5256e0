"""ctypes binding of oracle/librt_oracle.so (the CPU restatement; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")
_LIB = None


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays", "shadow_rays", "node_tests", "leaf_visits", "tri_tests", "sphere_tests",
        "ub_leave_level0", "ub_level_overflow", "ub_stack_overflow")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Info(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "depth", "n_tris", "n_spheres", "n_lights", "n_mats", "n_nodes", "n_leaves",
        "n_empty_leaves", "n_refs", "max_depth", "max_leaf")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "librt_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(path)
        L.oracle_scene_load.restype = C.c_void_p
        L.oracle_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
        L.oracle_scene_free.argtypes = [C.c_void_p]
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_scene_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        L.oracle_dump_tree.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_dump_tris.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_dump_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_render_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(Counters), C.c_int]
        for f in ("oracle_intersect_triangle", "oracle_intersect_sphere"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            getattr(L, f).restype = C.c_int
        L.oracle_node_accept.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_node_accept.restype = C.c_int
        L.oracle_tri_overlaps_box.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_tri_overlaps_box.restype = C.c_int
        L.oracle_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.oracle_provide_texture.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def provide_texture(file_path, rgb):
    """Hands the checker the decoded pixels ((H, W, 3) uint8) of the texture the resources call `file_path`."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    lib().oracle_provide_texture(file_path.encode(), rgb.shape[1], rgb.shape[0], rgb.ctypes.data)


class OracleScene:
    """One parsed scene (camera, materials, lights, primitives, octree) held by the oracle."""

    def __init__(self, scene, width=512, height=512, depth=3, resources=None, base_dir=None):
        L = lib()
        scene_path = scene if os.path.isabs(scene) else os.path.join(ASSETS, scene)
        res = resources or os.path.join(ASSETS, "resources.json")
        base = base_dir if base_dir is not None else ASSETS
        self.h = L.oracle_scene_load(scene_path.encode(), res.encode(), width, height, depth, base.encode())
        if not self.h:
            raise RuntimeError("oracle: " + L.oracle_last_error().decode())
        self.W, self.H = width, height

    def close(self):
        if self.h:
            lib().oracle_scene_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def info(self):
        i = Info()
        lib().oracle_scene_info(self.h, C.byref(i))
        return i.as_dict()

    def tree(self):
        i = self.info()
        kind = np.zeros(i["n_nodes"], np.int32)
        bounds = np.zeros((i["n_nodes"], 6), np.float32)
        refs = np.zeros(max(i["n_refs"], 1), np.int32)
        lib().oracle_dump_tree(self.h, kind.ctypes.data, bounds.ctypes.data, refs.ctypes.data)
        return kind, bounds, refs[:i["n_refs"]]

    def tris(self):
        n = self.info()["n_tris"]
        v = np.zeros((max(n, 1), 9), np.float32)
        nn = np.zeros((max(n, 1), 9), np.float32)
        m = np.zeros(max(n, 1), np.int32)
        lib().oracle_dump_tris(self.h, v.ctypes.data, nn.ctypes.data, m.ctypes.data)
        return v[:n], nn[:n], m[:n]

    def camera(self):
        out = np.zeros(19, np.float32)
        lib().oracle_dump_camera(self.h, out.ctypes.data)
        return out

    def render(self, row_begin=0, row_end=None, threads=1, want_ids=True):
        row_end = self.H if row_end is None else row_end
        rgb = np.zeros((self.H, self.W, 3), np.uint8)
        ids = np.full((self.H, self.W), -1, np.int32) if want_ids else None
        ts = np.zeros((self.H, self.W), np.float32) if want_ids else None
        ct = Counters()
        lib().oracle_render_rows(self.h, row_begin, row_end, rgb.ctypes.data,
                                 ids.ctypes.data if want_ids else None, ts.ctypes.data if want_ids else None,
                                 C.byref(ct), threads)
        return rgb, ids, ts, ct.as_dict()
