"""rt_scene_create: a scene handed over as flat arrays in memory (what the reference's own parse() + host model hold after
main.cu:18-23) must be the scene rt_scene_load makes of the files - same octree, same records, and on the GPU the same bytes."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as rt


def _own_copy(d):
    """The description again with every array copied into memory of the test's own (then handed to rt_scene_create and thrown
    away before the scene is used: the library must have copied what it needs)."""
    keep = []

    def arr(ptr, n, ctype):
        if n == 0 or not ptr:
            return C.cast(None, C.POINTER(ctype))
        buf = (ctype * n)()
        C.memmove(buf, ptr, C.sizeof(ctype) * n)
        keep.append(buf)
        return C.cast(buf, C.POINTER(ctype))

    o = rt.SceneDesc()
    C.memmove(C.byref(o.camera), C.byref(d.camera), C.sizeof(rt.CameraDesc))
    for n in ("n_lights", "n_materials", "n_mat_index", "n_spheres", "n_meshes", "n_triangles"):
        setattr(o, n, getattr(d, n))
    o.lights = arr(d.lights, d.n_lights, rt.LightDesc)
    o.materials = arr(d.materials, d.n_materials, rt.MaterialDesc)
    o.mat_index = arr(d.mat_index, d.n_mat_index, C.c_int32)
    o.spheres = arr(d.spheres, d.n_spheres, rt.SphereDesc)
    o.mesh_mat_first = arr(d.mesh_mat_first, d.n_meshes, C.c_int32)
    o.mesh_mat_count = arr(d.mesh_mat_count, d.n_meshes, C.c_int32)
    o.tri_verts = arr(d.tri_verts, 9 * d.n_triangles, C.c_float)
    o.tri_normals = arr(d.tri_normals, 9 * d.n_triangles, C.c_float)
    o.tri_uv = arr(d.tri_uv, 6 * d.n_triangles, C.c_float)
    o.tri_mesh = arr(d.tri_mesh, d.n_triangles, C.c_int32)
    return o, keep


def _recreate(name, w, h, depth=3):
    a = rt.Scene("scenes/%s.json" % name, w, h, depth)
    desc, keep = _own_copy(a.describe())
    b = rt.Scene.from_desc(desc)
    for buf in keep:                      # scribble over the caller's arrays: the scene must not be looking at them any more
        C.memset(buf, 0xff, C.sizeof(buf))
    del keep
    return a, b


@pytest.mark.parametrize("name", ["meshMedBunny", "heliumAttenuatedSphere", "boxedSpheres", "meshTransparentBunny"])
def test_scene_from_arrays_is_the_scene_from_files(name):
    a, b = _recreate(name, 160, 120)
    ia, ib = a.info(), b.info()
    for k in ia:
        if not k.endswith("_seconds"):
            assert ia[k] == ib[k], (k, ia[k], ib[k])
    for x, y in zip(a.tree(), b.tree()):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
    for x, y in zip(a.triangles(), b.triangles()):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
    assert np.array_equal(a.camera().view(np.uint32), b.camera().view(np.uint32))
    da, db = a.describe(), b.describe()
    assert bytes(da.camera) == bytes(db.camera)
    for n, t in (("lights", "n_lights"), ("materials", "n_materials"), ("spheres", "n_spheres")):
        cnt = getattr(da, t)
        assert cnt == getattr(db, t)
        for i in range(cnt):
            assert bytes(getattr(da, n)[i]) == bytes(getattr(db, n)[i]), (n, i)


def test_scene_from_arrays_refuses_bad_descriptions():
    a = rt.Scene("scenes/meshMedBunny.json", 64, 64, 3)
    d, keep = _own_copy(a.describe())
    d.mat_index[0] = d.n_materials                    # a material that does not exist
    with pytest.raises(rt.RtError) as e:
        rt.Scene.from_desc(d)
    assert e.value.status == -1
    d, keep = _own_copy(a.describe())
    d.tri_mesh[5] = d.n_meshes
    with pytest.raises(rt.RtError) as e:
        rt.Scene.from_desc(d)
    assert e.value.status == -1
    d, keep = _own_copy(a.describe())
    d.materials[0].type = 4                            # DMatType TEXTURE: the reference has no behaviour for it
    with pytest.raises(rt.RtError) as e:
        rt.Scene.from_desc(d)
    assert e.value.status == -1 and "TEXTURE" in str(e.value)
    d, keep = _own_copy(a.describe())
    d.camera.ray_depth = 10
    with pytest.raises(rt.RtError) as e:
        rt.Scene.from_desc(d)
    assert e.value.status == -4
    d, keep = _own_copy(a.describe())
    d.n_lights = 3
    d.lights = C.cast(None, C.POINTER(rt.LightDesc))
    with pytest.raises(rt.RtError):
        rt.Scene.from_desc(d)


def test_scene_from_hand_made_arrays():
    # no file involved at all: one triangle over one sphere, one light
    d = rt.SceneDesc()
    cam = rt.Scene("scenes/sphere.json", 32, 32, 3).describe().camera
    C.memmove(C.byref(d.camera), C.byref(cam), C.sizeof(rt.CameraDesc))
    lights = (rt.LightDesc * 1)(); lights[0].col[:] = [1, 1, 1]; lights[0].pos[:] = [0, 5, 5]; lights[0].type = 0
    mats = (rt.MaterialDesc * 1)(); mats[0].col[:] = [1, 0, 0]; mats[0].type = 0
    idx = (C.c_int32 * 2)(0, 0)
    sph = (rt.SphereDesc * 1)(); sph[0].c[:] = [0, 0, 0]; sph[0].r = 1.0; sph[0].mat_first = 0; sph[0].mat_count = 1
    verts = (C.c_float * 9)(-3, -1, -3, 3, -1, -3, 0, -1, 3)
    nrm = (C.c_float * 9)(0, 1, 0, 0, 1, 0, 0, 1, 0)
    tm = (C.c_int32 * 1)(0)
    mf, mc = (C.c_int32 * 1)(1), (C.c_int32 * 1)(1)
    d.n_lights, d.n_materials, d.n_mat_index, d.n_spheres, d.n_meshes, d.n_triangles = 1, 1, 2, 1, 1, 1
    d.lights, d.materials, d.mat_index, d.spheres = lights, mats, idx, sph
    d.mesh_mat_first, d.mesh_mat_count, d.tri_verts, d.tri_normals, d.tri_mesh = mf, mc, verts, nrm, tm
    s = rt.Scene.from_desc(d)
    i = s.info()
    assert i["n_triangles"] == 1 and i["n_spheres"] == 1 and i["n_lights"] == 1 and i["n_nodes"] == 9 and i["n_leaf_refs"] >= 1


@pytest.mark.parametrize("bad", ["nan", "fan", "huge"])
def test_mesh_that_no_subdivision_separates_is_refused_at_once(bad):
    # 4000 triangles that every octant keeps whole: vertices that are not numbers, a fan through one point, or triangles as large as the root box.  Eight children per level, each with the full
    # set, would be 8^19 nodes; the build has to stop at the first branch that reaches the depth limit.
    import time
    n = 4000
    rng = np.random.default_rng(5)
    v = rng.uniform(-1, 1, (n, 9)).astype(np.float32)
    if bad == "nan":
        v[:, 0] = np.nan
    elif bad == "fan":
        v[:, 0:3] = 0.25
    else:
        v *= np.float32(1e30)
    d = rt.SceneDesc()
    cam = rt.Scene("scenes/sphere.json", 32, 32, 3).describe().camera
    C.memmove(C.byref(d.camera), C.byref(cam), C.sizeof(rt.CameraDesc))
    lights = (rt.LightDesc * 1)(); lights[0].col[:] = [1, 1, 1]; lights[0].pos[:] = [0, 5, 5]
    mats = (rt.MaterialDesc * 1)(); mats[0].col[:] = [1, 0, 0]
    idx = (C.c_int32 * 1)(0)
    verts = (C.c_float * (9 * n))(*v.ravel().tolist())
    nrm = (C.c_float * (9 * n))(*([0.0, 1.0, 0.0] * (3 * n)))
    tm = (C.c_int32 * n)()
    mf, mc = (C.c_int32 * 1)(0), (C.c_int32 * 1)(1)
    d.n_lights, d.n_materials, d.n_mat_index, d.n_spheres, d.n_meshes, d.n_triangles = 1, 1, 1, 0, 1, n
    d.lights, d.materials, d.mat_index = lights, mats, idx
    d.mesh_mat_first, d.mesh_mat_count, d.tri_verts, d.tri_normals, d.tri_mesh = mf, mc, verts, nrm, tm
    t0 = time.time()
    if bad == "nan":     # (no box keeps such a triangle: the scene is made, with nothing in its leaves)
        assert rt.Scene.from_desc(d).info()["n_leaf_refs"] == 0
    else:
        with pytest.raises(rt.RtError) as e:
            rt.Scene.from_desc(d)
        assert e.value.status == -4
    assert time.time() - t0 < 20


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h", [("meshMedBunny", 256, 192), ("heliumAttenuatedSphere", 256, 192), ("meshTransparentBunny", 200, 136),
                                      ("meshDragon", 640, 360)])
def test_scene_from_arrays_renders_the_same_bytes(name, w, h):
    a, b = _recreate(name, w, h)
    ra = rt.DeviceScene(a).render_host(flags=rt.RT_RENDER_COUNT_TESTS)
    rb = rt.DeviceScene(b).render_host(flags=rt.RT_RENDER_COUNT_TESTS)
    assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2].view(np.uint32), rb[2].view(np.uint32))
    for k in ("rays", "shadow_rays", "node_tests", "leaf_visits", "tri_tests", "sphere_tests"):
        assert ra[3][k] == rb[3][k], (k, ra[3][k], rb[3][k])
    pa = rt.DeviceScene(a).render_host()     # and the production variant
    pb = rt.DeviceScene(b).render_host()
    assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1])
