"""raytracer_amd -- host-side mirror of the render path's C ABI (include/rt_amd.h).

The package is plumbing only: it loads ``librt_amd.so`` (HIP kernels for gfx950 + the C++ scene
builder) with ctypes and exposes the same contract the reference exposes through its command line:
a scene JSON + resources JSON + size + max depth go in, a top-left-origin RGB8 image comes out
(reference: ``run -s scene.json [-w W -h H -m depth -r resources.json -i image.png]``,
parsing/parsing.cu:145-175, main.cu:14-70).

There is no CPU rendering path in here: without the HIP library, or without a GPU, rendering raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
ASSETS = os.path.join(ROOT, "assets")
# RT_AMD_LIB selects another build of the same library (kernel tuning experiments).
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")

RT_RENDER_COUNT_TESTS = 1
RT_RENDER_TIME = 2
RT_RENDER_LITERAL = 4
RT_RENDER_STACKED = 8
RT_RENDER_QUEUED = 16
RT_RENDER_STRIPS = 32
RT_RENDER_EVERY_TRIANGLE = 64
RT_FRAME_ON_DEVICE = 128

# Every symbol include/rt_amd.h declares.
ABI_SYMBOLS = (
    "rt_scene_load", "rt_scene_create", "rt_scene_describe", "rt_scene_destroy", "rt_scene_get_info", "rt_scene_dump_tree", "rt_scene_dump_triangles",
    "rt_scene_dump_camera", "rt_device_prepare", "rt_scene_upload", "rt_dscene_destroy", "rt_dscene_fork", "rt_render_rows", "rt_render_rows_static", "rt_render_rows_host",
    "rt_host_alloc", "rt_host_free", "rt_band_count", "rt_render_bands", "rt_dscene_get_info",
    "rt_comm_unique_id", "rt_comm_init_rank", "rt_comm_init_all", "rt_comm_destroy", "rt_comm_group_begin", "rt_comm_group_end",
    "rt_gather_bands", "rt_unpack_bands", "rt_render_frame_gathered",
    "rt_frame_ctx_create", "rt_frame_ctx_destroy", "rt_frame_enqueue", "rt_frame_wait", "rt_frame_device_image",
    "rt_write_png", "rt_image_load", "rt_strerror", "rt_last_error", "rt_version",
)


class RtError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__("rt_amd status %d: %s" % (status, detail))
        self.status = status


class SceneInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "max_depth", "n_triangles", "n_spheres", "n_lights", "n_materials", "n_nodes", "n_leaves",
        "n_empty_leaves", "n_leaf_refs", "tree_depth", "max_leaf_size", "has_bounce_materials", "shadows_enabled")]
    _fields_ += [("_pad", C.c_int32), ("build_seconds", C.c_double), ("parse_seconds", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "_pad"}


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays", "shadow_rays", "node_tests", "leaf_visits", "tri_tests", "sphere_tests", "ub_leave_level0",
        "ub_level_overflow", "ub_stack_overflow", "rays_traced")] + [("wave_paths", C.c_uint64 * 8), ("kernel_ms", C.c_float), ("copy_ms", C.c_float),
                                                                    ("from_queues", C.c_uint32), ("queue_fallback", C.c_uint32)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "wave_paths"}
        d["wave_paths"] = list(self.wave_paths)
        return d


class CameraDesc(C.Structure):
    """rt_camera_desc: the reference's DCamera (camera/camera.cuh:10-17)."""
    _fields_ = [("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3), ("pos", C.c_float * 3),
                ("left", C.c_float), ("right", C.c_float), ("bottom", C.c_float), ("top", C.c_float),
                ("scene_refr", C.c_float), ("bg_col", C.c_float * 3),
                ("width", C.c_int32), ("height", C.c_int32), ("ray_depth", C.c_int32), ("shadows", C.c_int32)]


class LightDesc(C.Structure):
    _fields_ = [("col", C.c_float * 3), ("pos", C.c_float * 3), ("dir", C.c_float * 3), ("angle", C.c_float), ("type", C.c_int32)]


class MaterialDesc(C.Structure):
    _fields_ = [("col", C.c_float * 3), ("att", C.c_float * 3), ("param", C.c_float), ("type", C.c_int32)]


class SphereDesc(C.Structure):
    _fields_ = [("c", C.c_float * 3), ("r", C.c_float), ("mat_first", C.c_int32), ("mat_count", C.c_int32)]


class SceneDesc(C.Structure):
    """rt_scene_desc: a scene as flat arrays in the caller's memory (include/rt_amd.h)."""
    _fields_ = [("camera", CameraDesc)] + [(n, C.c_int32) for n in ("n_lights", "n_materials", "n_mat_index", "n_spheres", "n_meshes", "n_triangles")] + [
        ("lights", C.POINTER(LightDesc)), ("materials", C.POINTER(MaterialDesc)), ("mat_index", C.POINTER(C.c_int32)),
        ("spheres", C.POINTER(SphereDesc)), ("mesh_mat_first", C.POINTER(C.c_int32)), ("mesh_mat_count", C.POINTER(C.c_int32)),
        ("tri_verts", C.POINTER(C.c_float)), ("tri_normals", C.POINTER(C.c_float)), ("tri_uv", C.POINTER(C.c_float)),
        ("tri_mesh", C.POINTER(C.c_int32))]


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so (soname libamdhip64.so.7, the soname librt_amd.so
    asks for).  Two HIP runtimes in one process cannot both own the GPU, so when torch is installed its
    copy is loaded first and librt_amd.so binds to it; torch imported later finds the same file."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load librt_amd.so; raise loudly when it has not been built (there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `make` (or __graft_entry__.build()) first; "
                              "raytracer_amd has no non-HIP rendering path" % LIB_PATH)
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        L.rt_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.rt_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        L.rt_scene_describe.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.rt_scene_destroy.argtypes = [C.c_void_p]
        L.rt_scene_destroy.restype = None
        L.rt_scene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
        L.rt_scene_dump_tree.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_scene_dump_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_scene_dump_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_scene_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.rt_dscene_destroy.argtypes = [C.c_void_p]
        L.rt_dscene_destroy.restype = None
        L.rt_dscene_fork.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.rt_host_alloc.argtypes = [C.c_size_t]
        L.rt_host_alloc.restype = C.c_void_p
        L.rt_host_free.argtypes = [C.c_void_p]
        L.rt_host_free.restype = None
        L.rt_render_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_uint, C.POINTER(Stats)]
        L.rt_render_rows_static.argtypes = L.rt_render_rows.argtypes
        L.rt_render_rows_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint,
                                          C.POINTER(Stats)]
        L.rt_band_count.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.rt_render_bands.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint, C.POINTER(Stats)]
        L.rt_dscene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo), C.POINTER(C.c_int)]
        L.rt_comm_unique_id.argtypes = [C.c_void_p]
        L.rt_comm_init_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.rt_comm_init_all.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        L.rt_comm_destroy.argtypes = [C.c_void_p]
        L.rt_comm_destroy.restype = None
        L.rt_gather_bands.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rt_unpack_bands.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rt_render_frame_gathered.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_uint,
                                               C.POINTER(Stats)]
        L.rt_frame_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.rt_frame_ctx_destroy.argtypes = [C.c_void_p]
        L.rt_frame_ctx_destroy.restype = None
        L.rt_frame_enqueue.argtypes = [C.c_void_p, C.c_int, C.c_uint]
        L.rt_frame_wait.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_float)]
        L.rt_frame_device_image.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.rt_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.rt_image_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
        L.rt_strerror.argtypes = [C.c_int]
        L.rt_strerror.restype = C.c_char_p
        L.rt_last_error.restype = C.c_char_p
        L.rt_version.restype = C.c_char_p
        _lib = L
    return _lib


def _check(status):
    if status != 0:
        L = lib()
        detail = L.rt_last_error().decode() or L.rt_strerror(status).decode()
        raise RtError(status, detail)


def version():
    return lib().rt_version().decode()


KERNEL_SOURCES = ("rt_kernel_wave.hip", "rt_trace_asm.inc", "rt_math.h", "rt_kernels.h", "rt_internal.h", "rt_device.hip")


def kernel_source_sha256():
    """Digest of the sources the render kernels are built from; recorded with every committed counter profile so that
    a profile is only ever quoted for the code it was taken with."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def asset(path):
    """Resolve a path such as 'scenes/meshDragon.json' against the bundled assets directory."""
    return path if os.path.isabs(path) else os.path.join(ASSETS, path)


class Scene:
    """Parsed scene with its octree, on the host (rt_scene)."""

    def __init__(self, scene, width=512, height=512, max_depth=3, resources=None, base_dir=None):
        self.h = C.c_void_p()
        res = resources or os.path.join(ASSETS, "resources.json")
        base = ASSETS if base_dir is None else base_dir
        _check(lib().rt_scene_load(asset(scene).encode(), res.encode(), base.encode(), width, height, max_depth,
                                   C.byref(self.h)))
        self.width, self.height, self.max_depth = width, height, max_depth

    @classmethod
    def from_desc(cls, desc):
        """A scene from arrays in memory (rt_scene_create); the library copies what it needs."""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        _check(lib().rt_scene_create(C.byref(desc), C.byref(self.h)))
        self.width, self.height, self.max_depth = desc.camera.width, desc.camera.height, desc.camera.ray_depth
        return self

    def describe(self):
        """This scene as an rt_scene_desc whose pointers reach into the scene's own memory (valid while the scene lives)."""
        d = SceneDesc()
        _check(lib().rt_scene_describe(self.h, C.byref(d)))
        return d

    def close(self):
        if getattr(self, "h", None):
            lib().rt_scene_destroy(self.h)
            self.h = None

    __del__ = close

    def info(self):
        i = SceneInfo()
        _check(lib().rt_scene_get_info(self.h, C.byref(i)))
        return i.as_dict()

    def tree(self):
        i = self.info()
        kind = np.zeros(max(i["n_nodes"], 1), np.int32)
        bounds = np.zeros((max(i["n_nodes"], 1), 6), np.float32)
        refs = np.zeros(max(i["n_leaf_refs"], 1), np.int32)
        _check(lib().rt_scene_dump_tree(self.h, kind.ctypes.data, bounds.ctypes.data, refs.ctypes.data))
        return kind[:i["n_nodes"]], bounds[:i["n_nodes"]], refs[:i["n_leaf_refs"]]

    def triangles(self):
        n = self.info()["n_triangles"]
        v = np.zeros((max(n, 1), 9), np.float32)
        nn = np.zeros((max(n, 1), 9), np.float32)
        m = np.zeros(max(n, 1), np.int32)
        _check(lib().rt_scene_dump_triangles(self.h, v.ctypes.data, nn.ctypes.data, m.ctypes.data))
        return v[:n], nn[:n], m[:n]

    def camera(self):
        out = np.zeros(19, np.float32)
        _check(lib().rt_scene_dump_camera(self.h, out.ctypes.data))
        return out


class DeviceScene:
    """A Scene resident in one GPU's HBM (rt_dscene)."""

    def __init__(self, scene, device=0, _fork_of=None):
        self.h = C.c_void_p()
        self.scene = scene
        self.device = device
        self.width, self.height = scene.width, scene.height
        if _fork_of is not None:
            _check(lib().rt_dscene_fork(_fork_of.h, C.byref(self.h)))
        else:
            _check(lib().rt_scene_upload(scene.h, device, C.byref(self.h)))

    def fork(self):
        """Another render context (counters, queues) on the same uploaded scene: one per stream that renders concurrently."""
        return DeviceScene(self.scene, self.device, _fork_of=self)

    def close(self):
        if getattr(self, "h", None):
            lib().rt_dscene_destroy(self.h)
            self.h = None

    __del__ = close

    def render_host(self, row_begin=0, row_end=None, want_ids=True, flags=0):
        """Render rows into fresh numpy arrays (device->host copy included)."""
        row_end = self.height if row_end is None else row_end
        rgb = np.zeros((self.height, self.width, 3), np.uint8)
        ids = np.full((self.height, self.width), -1, np.int32) if want_ids else None
        ts = np.zeros((self.height, self.width), np.float32) if want_ids else None
        st = Stats()
        _check(lib().rt_render_rows_host(self.h, row_begin, row_end, rgb.ctypes.data,
                                         ids.ctypes.data if want_ids else None,
                                         ts.ctypes.data if want_ids else None, flags, C.byref(st)))
        return rgb, ids, ts, st.as_dict()

    def render_into(self, rgb_ptr, row_begin=0, row_end=None, ids_ptr=None, t_ptr=None, stream=None, flags=0,
                    want_stats=False):
        """Render rows into device buffers given as raw pointers (e.g. torch tensor .data_ptr())."""
        row_end = self.height if row_end is None else row_end
        st = Stats() if want_stats else None
        _check(lib().rt_render_rows(self.h, row_begin, row_end, rgb_ptr, ids_ptr, t_ptr, stream, flags,
                                    C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None


    def render_static_into(self, rgb_ptr, row_begin=0, row_end=None, ids_ptr=None, t_ptr=None, stream=None, flags=0,
                           want_stats=False):
        """render_into through the entry point that allocates nothing (scratch in the library's static device array)."""
        row_end = self.height if row_end is None else row_end
        st = Stats() if want_stats else None
        _check(lib().rt_render_rows_static(self.h, row_begin, row_end, rgb_ptr, ids_ptr, t_ptr, stream, flags,
                                           C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None

    def band_count(self, band_rows, rank, nranks):
        n = lib().rt_band_count(self.h, band_rows, rank, nranks)
        if n < 0:
            raise RtError(n, "rt_band_count: bad argument")
        return n

    def render_bands_into(self, rgb_ptr, band_rows, rank, nranks, ids_ptr=None, t_ptr=None, stream=None, flags=0,
                          want_stats=False):
        """Render this rank's row bands, packed, into a device buffer of band_count*band_rows rows."""
        st = Stats() if want_stats else None
        _check(lib().rt_render_bands(self.h, band_rows, rank, nranks, rgb_ptr, ids_ptr, t_ptr, stream, flags,
                                     C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None


COMM_ID_BYTES = 128


def comm_unique_id():
    """128 bytes that rank 0 makes and every rank of a communicator needs (send them over any channel)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib().rt_comm_unique_id(buf))
    return buf.raw


class Comm:
    """One rank of an RCCL communicator (rt_comm): the framebuffer gather of a frame rendered on several GPUs."""

    def __init__(self, handle, rank, nranks):
        self.h, self.rank, self.nranks = handle, rank, nranks

    @classmethod
    def init_rank(cls, unique_id, nranks, rank, device):
        h = C.c_void_p()
        _check(lib().rt_comm_init_rank(unique_id, nranks, rank, device, C.byref(h)))
        return cls(h, rank, nranks)

    @classmethod
    def init_all(cls, devices):
        n = len(devices)
        hs = (C.c_void_p * n)()
        _check(lib().rt_comm_init_all(n, (C.c_int * n)(*devices), hs))
        return [cls(C.c_void_p(hs[i]), i, n) for i in range(n)]

    def gather_bands(self, local_ptr, bytes_per_rank, gathered_ptr, stream=None):
        _check(lib().rt_gather_bands(self.h, local_ptr, bytes_per_rank, gathered_ptr, stream))

    def close(self):
        if getattr(self, "h", None):
            lib().rt_comm_destroy(self.h)
            self.h = None

    __del__ = close


def unpack_bands(gathered_ptr, bytes_per_rank, nranks, band_rows, width, height, bytes_per_pixel, image_ptr, stream=None):
    """Rank 0, on the device: the gathered packed tiles back into image order."""
    _check(lib().rt_unpack_bands(gathered_ptr, bytes_per_rank, nranks, band_rows, width, height, bytes_per_pixel, image_ptr, stream))


def render_frame_gathered(dscenes, comms, band_rows=16, flags=0):
    """One frame on the devices of `dscenes` (one process): bands, one RCCL gather, the frame in host memory."""
    n = len(dscenes)
    w, h = dscenes[0].width, dscenes[0].height
    rgb = np.zeros((h, w, 3), np.uint8)
    st = Stats()
    _check(lib().rt_render_frame_gathered((C.c_void_p * n)(*[d.h.value for d in dscenes]), (C.c_void_p * n)(*[c.h.value for c in comms]),
                                          n, band_rows, rgb.ctypes.data, flags, C.byref(st)))
    return rgb, st.as_dict()


class FramePipeline:
    """rt_frame_ctx: frames in flight on the devices of one process; everything is allocated here, nothing per frame."""

    def __init__(self, dscenes, comms=None, band_rows=16, slots=2):
        n = len(dscenes)
        self.h = C.c_void_p()
        self.width, self.height, self.slots = dscenes[0].width, dscenes[0].height, slots
        self._keep = (dscenes, comms)
        cs = (C.c_void_p * n)(*[c.h.value for c in comms]) if comms else None
        _check(lib().rt_frame_ctx_create((C.c_void_p * n)(*[d.h.value for d in dscenes]), cs, n, band_rows, slots, C.byref(self.h)))

    def enqueue(self, slot, flags=0):
        _check(lib().rt_frame_enqueue(self.h, slot, flags))

    def wait(self, slot, copy=True):
        """The slot's frame as an (H, W, 3) uint8 array (a copy unless copy=False: then a view of the slot's page-locked
        buffer, valid until the slot is enqueued again) and its first-launch-to-host time in ms."""
        p, ms = C.c_void_p(), C.c_float()
        _check(lib().rt_frame_wait(self.h, slot, C.byref(p), C.byref(ms)))
        if not p.value:                 # enqueued with RT_FRAME_ON_DEVICE: the frame stayed in device memory (device_image)
            return None, ms.value
        buf = (C.c_ubyte * (self.width * self.height * 3)).from_address(p.value)
        a = np.frombuffer(buf, np.uint8).reshape(self.height, self.width, 3)
        return (a.copy() if copy else a), ms.value

    def device_image(self, slot):
        """Device address of the slot's frame (W*H*3 bytes on the first device)."""
        p = C.c_void_p()
        _check(lib().rt_frame_device_image(self.h, slot, C.byref(p)))
        return p.value

    def close(self):
        if getattr(self, "h", None):
            lib().rt_frame_ctx_destroy(self.h)
            self.h = None

    __del__ = close


def band_rows_of(height, band_rows, rank, nranks):
    """Image rows owned by `rank` under the band partition, in packed order."""
    rows = []
    bands = (height + band_rows - 1) // band_rows
    for b in range(rank, bands, nranks):
        rows.extend(range(b * band_rows, min((b + 1) * band_rows, height)))
    return rows


def assemble_bands(tiles, height, band_rows):
    """Full image from the per-rank packed band buffers of one gather (tiles[r] = rank r's buffer)."""
    nranks = len(tiles)
    out = np.zeros((height,) + tuple(tiles[0].shape[1:]), tiles[0].dtype)
    for r, tile in enumerate(tiles):
        rows = band_rows_of(height, band_rows, r, nranks)
        out[rows] = np.asarray(tile)[:len(rows)]
    return out


def load_image(path):
    """A texture image (PNG or baseline JPEG) as the library decodes it: (H, W, 3) uint8, row 0 on top."""
    w, h = C.c_int(), C.c_int()
    _check(lib().rt_image_load(path.encode(), C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 3), np.uint8)
    _check(lib().rt_image_load(path.encode(), C.byref(w), C.byref(h), out.ctypes.data, out.nbytes))
    return out


def write_png(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    _check(lib().rt_write_png(path.encode(), w, h, rgb.ctypes.data))
