"""prt_amd -- Python host for the MI355X path-tracing hot path.

The product is libprt_hip.so (hand-written HIP kernels for gfx950 + the C++ host classes that
mirror the reference's Scene/Camera/Mesh/Bvh/Image/PathTracer surface).  This module is the thin
Python mirror of that surface over ctypes: same names, same argument meaning, used by the tests,
bench.py and __graft_entry__.  There is no CPU rendering path here: without the HIP library or a
GPU the render calls raise.

Reference citations (file:line under /root/reference/src) are in include/prt_hip.h and prt_host.h.
"""
import ctypes as C
import os

import numpy as np

from . import _build

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libprt_hip.so")


class PrtError(RuntimeError):
    pass


# ----------------------------------------------------------------------------- C structs (include/prt_hip.h)
class Material(C.Structure):
    _fields_ = [("diffuse", C.c_float * 3), ("emissive", C.c_float * 3), ("reflectionType", C.c_uint32),
                ("alphaTest", C.c_uint32), ("diffuseMap", C.c_int32), ("bumpMap", C.c_int32)]

    DIFFUSE, SPECULAR, REFRACTION = 0, 1, 2

    @classmethod
    def make(cls, diffuse=(0, 0, 0), emissive=(0, 0, 0), reflection=0):
        m = cls()
        m.diffuse[:] = [float(x) for x in diffuse]
        m.emissive[:] = [float(x) for x in emissive]
        m.reflectionType = reflection
        m.alphaTest = 0
        m.diffuseMap = -1
        m.bumpMap = -1
        return m


class BvhNode(C.Structure):
    _fields_ = [("lower", C.c_float * 3), ("upper", C.c_float * 3), ("primOrSecondNodeIndex", C.c_uint32),
                ("triVectorIndex", C.c_uint32), ("primCount", C.c_uint32), ("splitAxis", C.c_uint32)]


class MeshDesc(C.Structure):
    _fields_ = [("nodeCount", C.c_uint32), ("nodes", C.POINTER(BvhNode)), ("primCount", C.c_uint32),
                ("primRemapping", C.POINTER(C.c_uint32)), ("vertexCount", C.c_uint32), ("indices", C.POINTER(C.c_uint32)),
                ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("texcoords", C.POINTER(C.c_float)),
                ("materialCount", C.c_uint32), ("primMaterial", C.POINTER(C.c_uint32)), ("materials", C.POINTER(Material))]


class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("component", C.c_int32), ("texels", C.POINTER(C.c_uint8))]


class SceneDesc(C.Structure):
    _fields_ = [("meshCount", C.c_uint32), ("meshes", C.POINTER(MeshDesc)), ("textureCount", C.c_uint32),
                ("textures", C.POINTER(TextureDesc)), ("hasDirectionalLight", C.c_uint32), ("lightDir", C.c_float * 3),
                ("lightIntensity", C.c_float * 3), ("radius", C.c_float),
                ("hasInfiniteAreaLight", C.c_uint32), ("envWidth", C.c_int32), ("envHeight", C.c_int32),
                ("envTexels", C.POINTER(C.c_float)), ("envVerticalP", C.POINTER(C.c_float)), ("envHorizontalP", C.POINTER(C.c_float))]


class CameraDesc(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("width", C.c_uint32), ("height", C.c_uint32), ("invWidth", C.c_float), ("invHeight", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("samples", C.c_uint32), ("maxDepth", C.c_uint32), ("rrDepth", C.c_uint32), ("seed", C.c_uint32),
                ("exposure", C.c_float), ("tileSize", C.c_uint32), ("rank", C.c_uint32), ("nranks", C.c_uint32),
                ("countTraffic", C.c_uint32)]


class HipStats(C.Structure):
    _fields_ = ([(n, C.c_uint64) for n in ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx")]
                + [("modeBox", C.c_uint64 * 4), ("modeTri", C.c_uint64 * 4), ("modeTap", C.c_uint64 * 4), ("stackOverflow", C.c_uint64),
                   ("kernelMs", C.c_double), ("kernelMsSum", C.c_double), ("kernelLaunches", C.c_uint64)])

    def as_dict(self):
        out = {}
        for n, t in self._fields_:
            v = getattr(self, n)
            out[n] = [int(x) for x in v] if n.startswith("mode") else (float(v) if n.startswith("kernelMs") else int(v))
        return out


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("i", C.c_float), ("j", C.c_float), ("k", C.c_float), ("primId", C.c_uint32),
                ("meshId", C.c_uint32)]


HIT_DTYPE = np.dtype([("t", "<f4"), ("i", "<f4"), ("j", "<f4"), ("k", "<f4"), ("primId", "<u4"), ("meshId", "<u4")])
NODE_DTYPE = np.dtype([("lower", "<f4", 3), ("upper", "<f4", 3), ("primOrSecondNodeIndex", "<u4"),
                       ("triVectorIndex", "<u4"), ("primCount", "<u4"), ("splitAxis", "<u4")])
MATERIAL_DTYPE = np.dtype([("diffuse", "<f4", 3), ("emissive", "<f4", 3), ("reflectionType", "<u4"),
                           ("alphaTest", "<u4"), ("diffuseMap", "<i4"), ("bumpMap", "<i4")])

# every symbol include/prt_hip.h and include/prt_host.h declare
EXPORTS = [
    "prt_hip_device_count", "prt_hip_create", "prt_hip_destroy", "prt_hip_last_error", "prt_hip_source_sha16", "prt_hip_device_info",
    "prt_hip_upload_scene", "prt_hip_set_camera", "prt_hip_render", "prt_hip_render_gbuffer", "prt_hip_download", "prt_hip_framebuffer", "prt_hip_gather",
    "prt_hip_build_bvh", "prt_hip_comm_unique_id", "prt_hip_comm_init", "prt_hip_comm_adopt", "prt_hip_comm_destroy", "prt_hip_gather_rccl", "prt_hip_gather_payload_bytes",
    "prt_hip_get_stats",
    "prt_host_mesh_cornell", "prt_host_mesh_load_obj", "prt_host_mesh_from_arrays", "prt_host_mesh_displaced_sphere",
    "prt_host_mesh_atrium", "prt_host_mesh_destroy", "prt_host_mesh_transform", "prt_host_mesh_calculate_vertex_normals",
    "prt_host_mesh_calculate_bounds", "prt_host_mesh_prim_count", "prt_host_scene_create", "prt_host_scene_destroy",
    "prt_host_scene_add_mesh", "prt_host_scene_set_directional_light", "prt_host_scene_set_env_light", "prt_host_scene_load_env_light", "prt_host_scene_describe", "prt_host_scene_bbox", "prt_host_save_exr", "prt_host_save_ppm",
    "prt_host_camera_create", "prt_host_bvh_build", "prt_host_free",
]

# include/prt_hip_test.h: row-level entry points of the TEST build of the library (libprt_hip_test.so); the product does not export them
TEST_EXPORTS = ["prt_hip_trace_rays", "prt_hip_test_leaf", "prt_hip_test_sincos", "prt_hip_test_powf", "prt_hip_test_camera"]
TEST_LIB_PATH = os.path.join(_HERE, "lib", "libprt_hip_test.so")

_lib = None
_test_lib = None


def build(force=False):
    """Compile libprt_hip.so (the product) and libprt_hip_test.so (the same sources plus the row-level test entry points)
    for gfx950 with hipcc, in tree."""
    _build.build_library(force=force, test_entry_points=True)
    return _build.build_library(force=force)


def source_sha16():
    """Hash of the kernel sources in the tree (what a build now would be stamped with)."""
    return _build.source_sha16()


def loaded_source_sha16():
    """Hash of the kernel sources the LOADED library was built from (prt_hip_source_sha16): measurements are stamped with this."""
    return lib().prt_hip_source_sha16().decode()


def lib():
    """Load libprt_hip.so.  Fails loudly when it has not been built: the HIP library IS the product."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, False)
    return _lib


def test_lib():
    """Load libprt_hip_test.so: the product's sources built with -DPRT_TEST_ENTRY_POINTS (row-level parity tests only)."""
    global _test_lib
    if _test_lib is None:
        _test_lib = _load(TEST_LIB_PATH, True)
    return _test_lib


def _load(path, with_test_entry_points):
    if not os.path.exists(path):
        raise PrtError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950).  prt_amd has no CPU fallback.")
    L = C.CDLL(path)
    vp, f32p, u32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.prt_hip_last_error.restype = C.c_char_p
    L.prt_hip_source_sha16.restype = C.c_char_p
    L.prt_hip_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.prt_hip_destroy.argtypes = [vp]
    L.prt_hip_destroy.restype = None
    L.prt_hip_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    L.prt_hip_upload_scene.argtypes = [vp, C.POINTER(SceneDesc)]
    L.prt_hip_set_camera.argtypes = [vp, C.POINTER(CameraDesc)]
    L.prt_hip_render.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(RenderParams), vp, vp]
    L.prt_hip_download.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.prt_hip_render_gbuffer.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, vp, vp]
    L.prt_hip_gather.argtypes = [C.POINTER(vp), C.c_int, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.prt_hip_build_bvh.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp, vp, u32p, vp, C.POINTER(C.c_double)]
    L.prt_hip_comm_unique_id.argtypes = [vp]
    L.prt_hip_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.prt_hip_comm_adopt.argtypes = [vp, vp]
    L.prt_hip_comm_destroy.argtypes = [vp]
    L.prt_hip_gather_rccl.argtypes = [vp, vp, C.c_int, vp]
    L.prt_hip_gather_payload_bytes.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.prt_hip_framebuffer.restype = vp
    L.prt_hip_framebuffer.argtypes = [vp]
    L.prt_hip_get_stats.argtypes = [vp, C.POINTER(HipStats)]
    if with_test_entry_points:
        L.prt_hip_trace_rays.argtypes = [vp, C.c_int, C.c_uint32, vp, vp, C.c_float, vp]
        L.prt_hip_test_leaf.argtypes = [vp, C.c_uint32, vp, vp]
        L.prt_hip_test_sincos.argtypes = [vp, C.c_uint32, vp, vp, vp]
        L.prt_hip_test_powf.argtypes = [vp, C.c_uint32, vp, vp]
        L.prt_hip_test_camera.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    for n in ("prt_host_mesh_cornell", "prt_host_mesh_load_obj", "prt_host_mesh_from_arrays", "prt_host_mesh_displaced_sphere",
              "prt_host_mesh_atrium", "prt_host_scene_create"):
        getattr(L, n).restype = vp
    L.prt_host_mesh_cornell.argtypes = [C.c_int]
    L.prt_host_mesh_load_obj.argtypes = [C.c_char_p, C.POINTER(Material)]
    L.prt_host_mesh_from_arrays.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp]
    L.prt_host_mesh_displaced_sphere.argtypes = [C.c_uint32, C.c_float, f32p, C.POINTER(Material), C.c_uint32]
    L.prt_host_mesh_atrium.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_float]
    L.prt_host_mesh_destroy.argtypes = [vp]
    L.prt_host_mesh_destroy.restype = None
    L.prt_host_mesh_transform.argtypes = [vp, C.c_float, f32p]
    L.prt_host_mesh_transform.restype = None
    L.prt_host_mesh_calculate_vertex_normals.argtypes = [vp]
    L.prt_host_mesh_calculate_vertex_normals.restype = None
    L.prt_host_mesh_calculate_bounds.argtypes = [vp]
    L.prt_host_mesh_calculate_bounds.restype = None
    L.prt_host_mesh_prim_count.argtypes = [vp]
    L.prt_host_mesh_prim_count.restype = C.c_uint32
    L.prt_host_scene_destroy.argtypes = [vp]
    L.prt_host_scene_destroy.restype = None
    L.prt_host_scene_add_mesh.argtypes = [vp, vp]
    L.prt_host_scene_set_directional_light.argtypes = [vp, f32p, f32p]
    L.prt_host_scene_set_directional_light.restype = None
    L.prt_host_scene_set_env_light.argtypes = [vp, C.c_int32, C.c_int32, vp]
    L.prt_host_scene_set_env_light.restype = None
    L.prt_host_scene_load_env_light.argtypes = [vp, C.c_char_p]
    L.prt_host_save_exr.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp, C.c_int]
    L.prt_host_save_ppm.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp, C.c_int]
    L.prt_host_scene_describe.argtypes = [vp]
    L.prt_host_scene_describe.restype = C.POINTER(SceneDesc)
    L.prt_host_scene_bbox.argtypes = [vp, f32p]
    L.prt_host_scene_bbox.restype = None
    L.prt_host_camera_create.argtypes = [f32p, f32p, C.c_uint32, C.c_uint32, C.POINTER(CameraDesc)]
    L.prt_host_camera_create.restype = None
    L.prt_host_bvh_build.argtypes = [C.c_uint32, vp, vp, C.c_int, C.POINTER(C.POINTER(BvhNode)), u32p, C.POINTER(u32p)]
    L.prt_host_free.argtypes = [vp]
    L.prt_host_free.restype = None
    return L


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _check(rc, what):
    if rc != 0:
        raise PrtError(f"{what} failed ({rc}): {lib().prt_hip_last_error().decode()}")


# ----------------------------------------------------------------------------- host mirror
class Mesh:
    """prt::Mesh (mesh.h:30-105) before it is handed to a Bvh."""

    def __init__(self, handle):
        if not handle:
            raise PrtError("mesh creation failed")
        self._h = handle

    @classmethod
    def cornell_box(cls, box=True):
        return cls(lib().prt_host_mesh_cornell(int(box)))

    @classmethod
    def load_obj(cls, path, material=None):
        return cls(lib().prt_host_mesh_load_obj(os.fsencode(path), C.byref(material) if material is not None else None))

    @classmethod
    def from_arrays(cls, indices, positions, prim_material, materials, normals=None, texcoords=None):
        indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        prim_material = np.ascontiguousarray(prim_material, dtype=np.uint32)
        materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        n = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32)
        t = None if texcoords is None else np.ascontiguousarray(texcoords, dtype=np.float32)
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
        return cls(lib().prt_host_mesh_from_arrays(len(indices), len(positions), len(materials), p(indices), p(positions), p(n),
                                                   p(t), p(prim_material), p(materials)))

    @classmethod
    def displaced_sphere(cls, target_tris, radius, center, material, seed=1):
        return cls(lib().prt_host_mesh_displaced_sphere(target_tris, radius, _f3(center), C.byref(material), seed))

    @classmethod
    def atrium(cls, target_tris, seed=1, alpha_masked=True, bump_mapped=True, emissive_fraction=0.0):
        return cls(lib().prt_host_mesh_atrium(target_tris, seed, int(alpha_masked), int(bump_mapped), emissive_fraction))

    def transform(self, scale, translate):
        lib().prt_host_mesh_transform(self._h, scale, _f3(translate))

    def calculate_vertex_normals(self):
        lib().prt_host_mesh_calculate_vertex_normals(self._h)

    def calculate_bounds(self):
        lib().prt_host_mesh_calculate_bounds(self._h)

    @property
    def prim_count(self):
        return lib().prt_host_mesh_prim_count(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().prt_host_mesh_destroy(self._h)
            self._h = None


class Scene:
    """prt::Scene (scene.h:11-73) owning its Bvhs."""

    def __init__(self):
        self._h = lib().prt_host_scene_create()

    def add(self, mesh):
        """Bvh::build(std::move(mesh)) + Scene::add(bvh) (main.cpp:24-26)."""
        _check(lib().prt_host_scene_add_mesh(self._h, mesh._h), "prt_host_scene_add_mesh")
        mesh._h = None

    def set_directional_light(self, direction, intensity):
        lib().prt_host_scene_set_directional_light(self._h, _f3(direction), _f3(intensity))

    def set_infinite_area_light(self, env):
        """Scene::setInfiniteAreaLight (scene.h:42-45): `env` is a path to an OpenEXR file (scan lines; NO / RLE / ZIPS / ZIP) or an RGB PFM file, or a float array (h, w, 4) RGBA, row 0 = top."""
        if isinstance(env, (str, bytes, os.PathLike)):
            _check(lib().prt_host_scene_load_env_light(self._h, os.fsencode(env)), "prt_host_scene_load_env_light")
            return
        e = np.ascontiguousarray(env, dtype=np.float32)
        if e.ndim != 3 or e.shape[2] != 4:
            raise PrtError("environment map must be (height, width, 4) float RGBA")
        lib().prt_host_scene_set_env_light(self._h, e.shape[1], e.shape[0], e.ctypes.data_as(C.c_void_p))

    def describe(self):
        return lib().prt_host_scene_describe(self._h)

    def bbox(self):
        b = (C.c_float * 6)()
        lib().prt_host_scene_bbox(self._h, b)
        return np.array(b[:], dtype=np.float32)

    def arrays(self):
        """numpy copies of everything the descriptor points to (for checkers)."""
        d = self.describe().contents
        out = dict(meshes=[], textures=[], has_light=bool(d.hasDirectionalLight), light_dir=np.array(d.lightDir[:], dtype=np.float32),
                   light_intensity=np.array(d.lightIntensity[:], dtype=np.float32), radius=np.float32(d.radius))
        for i in range(d.meshCount):
            m = d.meshes[i]
            as_np = lambda p, n, dt: np.ctypeslib.as_array(p, shape=(n,)).view(dt).copy()  # noqa: E731
            out["meshes"].append(dict(
                nodes=np.frombuffer(C.string_at(m.nodes, m.nodeCount * C.sizeof(BvhNode)), dtype=NODE_DTYPE).copy(),
                remap=as_np(m.primRemapping, m.primCount, np.uint32),
                indices=as_np(m.indices, m.primCount * 3, np.uint32).reshape(-1, 3),
                positions=as_np(m.positions, m.vertexCount * 3, np.float32).reshape(-1, 3),
                normals=as_np(m.normals, m.vertexCount * 3, np.float32).reshape(-1, 3) if m.normals else None,
                texcoords=as_np(m.texcoords, m.vertexCount * 2, np.float32).reshape(-1, 2) if m.texcoords else None,
                prim_material=as_np(m.primMaterial, m.primCount, np.uint32),
                materials=np.frombuffer(C.string_at(m.materials, m.materialCount * C.sizeof(Material)), dtype=MATERIAL_DTYPE).copy()))
        for i in range(d.textureCount):
            t = d.textures[i]
            out["textures"].append(np.ctypeslib.as_array(t.texels, shape=(t.height, t.width, t.component)).copy())
        out["env"] = None
        if d.hasInfiniteAreaLight:
            w, h = d.envWidth, d.envHeight
            out["env"] = np.ctypeslib.as_array(d.envTexels, shape=(h, w, 4)).copy()
            out["env_vertical"] = np.ctypeslib.as_array(d.envVerticalP, shape=(h,)).copy()
            out["env_horizontal"] = np.ctypeslib.as_array(d.envHorizontalP, shape=(h * w,)).copy()
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().prt_host_scene_destroy(self._h)
            self._h = None


class Camera:
    """prt::Camera (camera.h:14-53)."""

    def __init__(self):
        self.desc = CameraDesc()

    def create(self, pos, direction, width, height):
        self.pos_arg = np.asarray(pos, dtype=np.float32)        # the arguments as given (the basis in desc is derived)
        self.dir_arg = np.asarray(direction, dtype=np.float32)
        lib().prt_host_camera_create(_f3(pos), _f3(direction), width, height, C.byref(self.desc))
        return self

    @property
    def width(self):
        return self.desc.width

    @property
    def height(self):
        return self.desc.height


class PathTracer:
    """prt::PathTracer (path_tracer.h:15-38) bound to one GPU through the C-ABI (include/prt_hip.h)."""

    def __init__(self, device=0, max_depth=14, rr_depth=4, seed=12345, test_entry_points=False):
        """test_entry_points=True binds the TEST build of the library (row-level entry points for the parity tests)."""
        L = self._L = test_lib() if test_entry_points else lib()
        self._row_level = test_entry_points
        self._ctx = C.c_void_p()
        self._chk(L.prt_hip_create(device, C.byref(self._ctx)), "prt_hip_create")
        self.max_depth, self.rr_depth, self.seed = max_depth, rr_depth, seed
        self._scene = None
        self._camera = None

    def _chk(self, rc, what):
        if rc != 0:
            raise PrtError(f"{what} failed ({rc}): {self._L.prt_hip_last_error().decode()}")

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.prt_hip_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int()
        self._chk(self._L.prt_hip_device_info(self._ctx, name, 256, C.byref(cus)), "prt_hip_device_info")
        return name.value.decode(), cus.value

    def upload_scene(self, scene):
        self._chk(self._L.prt_hip_upload_scene(self._ctx, scene.describe()), "prt_hip_upload_scene")
        self._scene = scene

    def set_camera(self, camera):
        self._chk(self._L.prt_hip_set_camera(self._ctx, C.byref(camera.desc)), "prt_hip_set_camera")
        self._camera = camera

    def params(self, samples, exposure=1.0, rank=0, nranks=1, count_traffic=False, max_depth=None, tile=16):
        p = RenderParams()
        p.samples = samples
        p.maxDepth = self.max_depth if max_depth is None else max_depth
        p.rrDepth = self.rr_depth
        p.seed = self.seed
        p.exposure = exposure
        p.tileSize = tile
        p.rank, p.nranks = rank, nranks
        p.countTraffic = int(count_traffic)
        return p

    def render_async(self, x0, y0, x1, y1, samples, d_rgb=None, stream=None, **kw):
        """PathTracer::TraceBlock on the GPU; d_rgb = device pointer (int) or None for the context framebuffer."""
        p = self.params(samples, **kw)
        self._chk(self._L.prt_hip_render(self._ctx, x0, y0, x1, y1, C.byref(p), d_rgb, stream), "prt_hip_render")

    def trace_block(self, x0, y0, x1, y1, samples, **kw):
        """Render the inclusive rectangle and return it as a (h, w, 3) float32 array."""
        self.render_async(x0, y0, x1, y1, samples, **kw)
        W, H = self._camera.width, self._camera.height
        img = np.zeros((H, W, 3), dtype=np.float32)
        self._download(img, x0, y0, x1, y1)
        self.last_stats = self.stats()  # raises on stack overflow
        return img[y0:y1 + 1, x0:x1 + 1].copy()

    def _download(self, img, x0, y0, x1, y1):
        """prt_hip_download reports an earlier launch's error (watchdog, stack overflow) WITHOUT clearing it -- the C-ABI's rule:
        only prt_hip_get_stats consumes it.  The calls of this class that download also own the frame, so they consume it
        here: the error is raised once and the next render on this tracer starts clean."""
        try:
            self._chk(self._L.prt_hip_download(self._ctx, img.ctypes.data_as(C.c_void_p), x0, y0, x1, y1), "prt_hip_download")
        except PrtError:
            try:
                self.stats()
            except PrtError:
                pass
            raise

    def gbuffer(self, kind, x0=0, y0=0, x1=None, y1=None, exposure=1.0):
        """GbufferVisualizer::TraceBlock (gbuffer_visualizer.cpp:17-51): kind 0 diffuse colour, 1 / 2 bump-mapped normal."""
        W, H = self._camera.width, self._camera.height
        x1 = W - 1 if x1 is None else x1
        y1 = H - 1 if y1 is None else y1
        self._chk(self._L.prt_hip_render_gbuffer(self._ctx, x0, y0, x1, y1, kind, self.seed, exposure, None, None), "prt_hip_render_gbuffer")
        img = np.zeros((H, W, 3), dtype=np.float32)
        self._download(img, x0, y0, x1, y1)
        return img[y0:y1 + 1, x0:x1 + 1].copy()

    def render(self, samples, **kw):
        W, H = self._camera.width, self._camera.height
        return self.trace_block(0, 0, W - 1, H - 1, samples, **kw)

    def build_bvh(self, indices, positions):
        """Bvh::build on the GPU (prt_hip_build_bvh): returns (nodes as NODE_DTYPE array, primRemapping, device ms)."""
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        n = len(idx)
        nodes = np.zeros(2 * n, dtype=NODE_DTYPE)
        remap = np.zeros(n, dtype=np.uint32)
        count, ms = C.c_uint32(), C.c_double()
        self._chk(self._L.prt_hip_build_bvh(self._ctx, n, idx.ctypes.data_as(C.c_void_p), len(pos), pos.ctypes.data_as(C.c_void_p),
                                            nodes.ctypes.data_as(C.c_void_p), C.byref(count), remap.ctypes.data_as(C.c_void_p), C.byref(ms)),
                  "prt_hip_build_bvh")
        return nodes[:count.value].copy(), remap, ms.value

    # ---- multi-process image gather over RCCL (include/prt_hip.h "image gather")
    def comm_init(self, unique_id, rank, nranks):
        """Collective: every rank calls it with rank 0's id (comm_unique_id(), shipped over any host channel)."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self._L.prt_hip_comm_init(self._ctx, buf, rank, nranks), "prt_hip_comm_init")

    def gather_rccl(self, d_rgb=None, root=0, stream=None):
        """Collective: moves the tiles every rank owns (1/nranks of the image) to `root` and de-interleaves them there."""
        self._chk(self._L.prt_hip_gather_rccl(self._ctx, d_rgb, root, stream), "prt_hip_gather_rccl")

    def gather_payload_bytes(self):
        n = C.c_uint64()
        self._chk(self._L.prt_hip_gather_payload_bytes(self._ctx, C.byref(n)), "prt_hip_gather_payload_bytes")
        return n.value

    def stats(self):
        st = HipStats()
        self._chk(self._L.prt_hip_get_stats(self._ctx, C.byref(st)), "prt_hip_get_stats")
        return st.as_dict()

    # ---- row-level entry points (parity tests; include/prt_hip_test.h, test build of the library only)
    def _need_row_level(self):
        if not self._row_level:
            raise PrtError("row-level entry points exist only in libprt_hip_test.so: PathTracer(test_entry_points=True)")

    def trace_rays(self, mode, org, dirs, max_t):
        self._need_row_level()
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        hits = np.zeros(len(org), dtype=HIT_DTYPE)
        self._chk(self._L.prt_hip_trace_rays(self._ctx, mode, len(org), org.ctypes.data_as(C.c_void_p), dirs.ctypes.data_as(C.c_void_p),
                                        max_t, hits.ctypes.data_as(C.c_void_p)), "prt_hip_trace_rays")
        return hits

    def test_leaf(self, records):
        self._need_row_level()
        rec = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 22)
        out = np.zeros((len(rec), 24), dtype=np.float32)
        self._chk(self._L.prt_hip_test_leaf(self._ctx, len(rec), rec.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), "prt_hip_test_leaf")
        return out

    def test_sincos(self, theta):
        self._need_row_level()
        th = np.ascontiguousarray(theta, dtype=np.float32)
        s, c = np.zeros_like(th), np.zeros_like(th)
        self._chk(self._L.prt_hip_test_sincos(self._ctx, len(th), th.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p),
                                         c.ctypes.data_as(C.c_void_p)), "prt_hip_test_sincos")
        return s, c

    def test_powf(self, x):
        self._need_row_level()
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.zeros_like(x)
        self._chk(self._L.prt_hip_test_powf(self._ctx, len(x), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)), "prt_hip_test_powf")
        return y

    def test_camera(self, x, y, state):
        self._need_row_level()
        out = np.zeros(92, dtype=np.float32)
        self._chk(self._L.prt_hip_test_camera(self._ctx, x, y, state, out.ctypes.data_as(C.c_void_p)), "prt_hip_test_camera")
        return out


def device_count():
    return lib().prt_hip_device_count()


# ----------------------------------------------------------------------------- the reference's scene setups (main.cpp:22-105)
def setup_cornell_box(width, height, teapot_obj=None, teapot_mesh=None):
    """setupCornellBox (main.cpp:22-55).  The teapot comes from an OBJ file or from a ready Mesh."""
    scene = Scene()
    scene.add(Mesh.cornell_box(True))
    if teapot_obj is not None:
        teapot_mesh = Mesh.load_obj(teapot_obj, Material.make(diffuse=(0.9, 0.9, 0.9), reflection=Material.SPECULAR))
        teapot_mesh.transform(0.005, (-0.5, 0.0, 0.5))
    if teapot_mesh is not None:
        teapot_mesh.calculate_vertex_normals()
        teapot_mesh.calculate_bounds()
        scene.add(teapot_mesh)
    camera = Camera().create((0, 0.965, 2.6), (0, 0, -1.0), width, height)
    return scene, camera, 1.0


def _normalize(v):
    v = np.asarray(v, dtype=np.float32)
    d = np.float32(v[0] * v[0]) + np.float32(v[1] * v[1]) + np.float32(v[2] * v[2])
    inv = np.float32(1.0) / np.sqrt(np.float32(d), dtype=np.float32)
    return (inv * v).astype(np.float32)


def setup_bunny_standin(width, height, tris=69451, seed=1):
    """BASELINE config 2 stand-in: Cornell box + a bunny-class displaced sphere (69 451 triangles
    asked for; a lat-long grid gives the nearest even count) in place of the teapot, plus the directional light of
    setupSanMiguelLowPoly (main.cpp:84) so that occlusion rays are exercised."""
    scene = Scene()
    scene.add(Mesh.cornell_box(True))
    m = Mesh.displaced_sphere(tris, 0.3, (-0.45, 0.36, 0.45), Material.make(diffuse=(0.8, 0.75, 0.7)), seed)
    m.calculate_vertex_normals()
    m.calculate_bounds()
    scene.add(m)
    scene.set_directional_light(_normalize((0.2, 1.0, 0.2)), (16.7, 15.6, 11.7))
    camera = Camera().create((0, 0.965, 2.6), (0, 0, -1.0), width, height)
    return scene, camera, 1.0


def setup_atrium_standin(width, height, tris=262000, seed=1, alpha=True, bump=True, emissive_fraction=0.0, light=True):
    """BASELINE config 3 stand-in: Sponza-class atrium, light as setupSponza (main.cpp:66)."""
    scene = Scene()
    m = Mesh.atrium(tris, seed, alpha, bump, emissive_fraction)
    m.calculate_vertex_normals()
    scene.add(m)
    if light:
        scene.set_directional_light(_normalize((0.05, 1.0, 0.1)), (16.7, 15.6, 11.7))
    camera = Camera().create((-15.0, 4.0, 0.5), (1.0, 0.08, -0.05), width, height)
    return scene, camera, 1.0


# ----------------------------------------------------------------------------- multi-GPU sharding (one process per GPU)
def owned_pixel_mask(width, height, rank, nranks, tile=16):
    """Boolean (height, width) mask of the pixels rank `rank` renders: 16x16 tiles (main.cpp:123-124) dealt round-robin,
    tile id = ty * tiles_per_row + tx, owner = id % nranks -- the rule prt_hip_render applies on the device."""
    ty, tx = np.meshgrid(np.arange(height) // tile, np.arange(width) // tile, indexing="ij")
    tiles_x = (width + tile - 1) // tile
    return ((ty * tiles_x + tx) % nranks) == rank


def save_exr(path, rgb, zip=True):
    """Image::saveExr (image.cpp:82-139): (H, W, 3) float image -> half-float OpenEXR with channels B, G, R (ZIP blocks like
    tinyexr's default, or raw scan lines)."""
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    _check(lib().prt_host_save_exr(os.fsencode(path), a.shape[1], a.shape[0], a.ctypes.data_as(C.c_void_p), int(zip)), "prt_host_save_exr")


def save_ppm(path, rgb, tonemap=True):
    """Image::savePpm (image.cpp:52-80): c/(c+1) tone map (optional), clamp, gamma 1/2.2, 8 bits."""
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    _check(lib().prt_host_save_ppm(os.fsencode(path), a.shape[1], a.shape[0], a.ctypes.data_as(C.c_void_p), int(tonemap)), "prt_host_save_ppm")


def comm_unique_id():
    """128 bytes naming a new RCCL communicator (rank 0 calls this and broadcasts the bytes)."""
    buf = C.create_string_buffer(128)
    _check(lib().prt_hip_comm_unique_id(buf), "prt_hip_comm_unique_id")
    return buf.raw


def gather_contexts(tracers, x0, y0, x1, y1):
    """prt_hip_gather: one process driving several contexts (tracer i rendered the rectangle with rank=i, nranks=len(tracers)
    into its own framebuffer); returns the assembled (H, W, 3) image."""
    cam = tracers[0]._camera
    img = np.zeros((cam.height, cam.width, 3), dtype=np.float32)
    arr = (C.c_void_p * len(tracers))(*[t._ctx for t in tracers])
    _check(lib().prt_hip_gather(arr, len(tracers), img.ctypes.data_as(C.c_void_p), x0, y0, x1, y1), "prt_hip_gather")
    return img


def owned_tile_ids(width, height, rank, nranks, tile=16):
    """Tile ids rank, rank + nranks, ... of the image's tile grid: the order of the tiles in a rank's packed buffer."""
    total = ((width + tile - 1) // tile) * ((height + tile - 1) // tile)
    return np.arange(rank, total, nranks, dtype=np.int64)


def pack_tiles(image, rank, nranks, tile=16):
    """Host mirror of the library's pack kernel (prt_gather.hip): the tiles `rank` owns, tile-major, (n, tile, tile, 3);
    pixels beyond the image's edge are zero."""
    H, W, _ = image.shape
    tiles_x = (W + tile - 1) // tile
    ids = owned_tile_ids(W, H, rank, nranks, tile)
    out = np.zeros((len(ids), tile, tile, 3), dtype=np.float32)
    for q, t in enumerate(ids):
        x0, y0 = int(t % tiles_x) * tile, int(t // tiles_x) * tile
        crop = image[y0:y0 + tile, x0:x0 + tile]
        out[q, :crop.shape[0], :crop.shape[1]] = crop
    return out


def unpack_tiles(image, packed, rank, nranks, tile=16):
    """Host mirror of the de-interleave kernel: writes rank's packed tiles into `image` (in place)."""
    H, W, _ = image.shape
    tiles_x = (W + tile - 1) // tile
    for q, t in enumerate(owned_tile_ids(W, H, rank, nranks, tile)):
        x0, y0 = int(t % tiles_x) * tile, int(t // tiles_x) * tile
        h, w = min(tile, H - y0), min(tile, W - x0)
        image[y0:y0 + h, x0:x0 + w] = packed[q, :h, :w]
    return image


def gather_image_host(framebuffer, rank, nranks, dst=0, tile=16):
    """The gather's data movement on HOST tensors over torch.distributed point-to-point (gloo in the CPU tests): every rank
    sends the tiles it owns, packed as the device packs them, to `dst`, which de-interleaves them -- the same ownership
    rule, tile order and payload as prt_hip_gather_rccl, without the GPU.  `framebuffer` is a (H, W, 3) float32 numpy array."""
    import torch
    import torch.distributed as dist
    H, W, _ = framebuffer.shape
    if rank != dst:
        mine = pack_tiles(framebuffer, rank, nranks, tile)
        if mine.size:
            dist.send(torch.from_numpy(mine), dst=dst)
        return framebuffer
    for r in range(nranks):
        n = len(owned_tile_ids(W, H, r, nranks, tile))
        if r == dst or n == 0:
            continue
        buf = torch.empty((n, tile, tile, 3), dtype=torch.float32)
        dist.recv(buf, src=r)
        unpack_tiles(framebuffer, buf.numpy(), r, nranks, tile)
    return framebuffer
