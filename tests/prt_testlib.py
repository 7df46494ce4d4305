"""Shared helpers for the test-suite: ctypes bindings of the oracle (checker only), the scene
description that both the oracle and the product consume, the PRTS scene-file writer for the
reference harness, and locations of built artefacts.

Nothing here is product code; the product is `prt_amd` (HIP kernels behind include/prt_hip.h).
"""
import ctypes as C
import os
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")


# ----------------------------------------------------------------------------- oracle bindings
class OrcMaterial(C.Structure):
    _fields_ = [("diffuse", C.c_float * 3), ("emissive", C.c_float * 3), ("reflectionType", C.c_uint32),
                ("alphaTest", C.c_uint32), ("diffuseMap", C.c_int32), ("bumpMap", C.c_int32)]


class OrcHit(C.Structure):
    _fields_ = [("t", C.c_float), ("i", C.c_float), ("j", C.c_float), ("k", C.c_float),
                ("primId", C.c_uint32), ("meshId", C.c_uint32)]


class OrcNode(C.Structure):
    _fields_ = [("lower", C.c_float * 3), ("upper", C.c_float * 3), ("primOrSecondNodeIndex", C.c_uint32),
                ("triVectorIndex", C.c_uint32), ("primCount", C.c_uint32), ("splitAxis", C.c_uint32)]


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("raysTraced", "occludedTraced", "nBox", "nTri", "nHit", "nTap", "nPx", "rngDraws")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class OrcCamera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("width", C.c_uint32), ("height", C.c_uint32), ("invWidth", C.c_float), ("invHeight", C.c_float)]


MATERIAL_DTYPE = np.dtype([("diffuse", "<f4", 3), ("emissive", "<f4", 3), ("reflectionType", "<u4"),
                           ("alphaTest", "<u4"), ("diffuseMap", "<i4"), ("bumpMap", "<i4")])
NODE_DTYPE = np.dtype([("lower", "<f4", 3), ("upper", "<f4", 3), ("primOrSecondNodeIndex", "<u4"),
                       ("triVectorIndex", "<u4"), ("primCount", "<u4"), ("splitAxis", "<u4")])
HIT_DTYPE = np.dtype([("t", "<f4"), ("i", "<f4"), ("j", "<f4"), ("k", "<f4"), ("primId", "<u4"), ("meshId", "<u4")])

_oracle = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


def oracle():
    """Load oracle/liboracle.so (building it with gcc if needed)."""
    global _oracle
    if _oracle is not None:
        return _oracle
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "prt_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build_oracle()
    L = C.CDLL(path)
    vp, f32p, u32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.orc_mesh_create.restype = vp
    L.orc_mesh_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp]
    L.orc_mesh_destroy.argtypes = [vp]
    L.orc_mesh_calculate_vertex_normals.argtypes = [vp]
    L.orc_mesh_calculate_bounds.argtypes = [vp]
    L.orc_mesh_normals.restype = f32p
    L.orc_mesh_normals.argtypes = [vp]
    L.orc_mesh_bbox.restype = f32p
    L.orc_mesh_bbox.argtypes = [vp]
    L.orc_bvh_build.restype = vp
    L.orc_bvh_build.argtypes = [vp]
    L.orc_bvh_destroy.argtypes = [vp]
    for n in ("orc_bvh_node_count", "orc_bvh_leaf_count", "orc_bvh_prim_count"):
        getattr(L, n).restype = C.c_uint32
        getattr(L, n).argtypes = [vp]
    L.orc_bvh_nodes.restype = C.POINTER(OrcNode)
    L.orc_bvh_nodes.argtypes = [vp]
    L.orc_bvh_prim_remap.restype = u32p
    L.orc_bvh_prim_remap.argtypes = [vp]
    L.orc_scene_create.restype = vp
    L.orc_scene_destroy.argtypes = [vp]
    L.orc_scene_add.argtypes = [vp, vp]
    L.orc_scene_set_directional_light.argtypes = [vp, f32p, f32p]
    L.orc_scene_add_texture.restype = C.c_int32
    L.orc_scene_add_texture.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp]
    L.orc_scene_radius.restype = C.c_float
    L.orc_scene_radius.argtypes = [vp]
    L.orc_scene_bbox.restype = f32p
    L.orc_scene_bbox.argtypes = [vp]
    L.orc_camera_create.argtypes = [C.POINTER(OrcCamera), f32p, f32p, C.c_uint32, C.c_uint32]
    L.orc_pixel_seed.restype = C.c_uint32
    L.orc_pixel_seed.argtypes = [C.c_uint32] * 4
    L.orc_rng_next.restype = C.c_uint32
    L.orc_rng_next.argtypes = [u32p]
    L.orc_rng_float.restype = C.c_float
    L.orc_rng_float.argtypes = [u32p]
    L.orc_intersect_triangle.restype = C.c_float
    L.orc_intersect_triangle.argtypes = [f32p, f32p, C.c_int, C.c_int, f32p, f32p, f32p, f32p]
    L.orc_intersect_triangle_scalar.restype = C.c_float
    L.orc_intersect_triangle_scalar.argtypes = [f32p, f32p, f32p, f32p, f32p, f32p]
    L.orc_bbox_intersect_t.restype = C.c_float
    L.orc_bbox_intersect_t.argtypes = [f32p, f32p, f32p, f32p]
    L.orc_bbox_intersect_bool.restype = C.c_int
    L.orc_bbox_intersect_bool.argtypes = [f32p, f32p, f32p, f32p, C.c_float]
    L.orc_bbox_intersect_soa.restype = C.c_int
    L.orc_bbox_intersect_soa.argtypes = [f32p, f32p, f32p, f32p, C.c_float]
    L.orc_ray_prepare_single.argtypes = [f32p, f32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_ray_prepare_soa.argtypes = [f32p, f32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_intersect_single.argtypes = [vp, f32p, f32p, C.c_float, C.POINTER(OrcHit), C.POINTER(OrcStats)]
    L.orc_intersect_packet.argtypes = [vp, vp, vp, f32p, C.c_float, vp, C.POINTER(OrcStats)]
    L.orc_occluded_single.restype = C.c_int
    L.orc_occluded_single.argtypes = [vp, f32p, f32p, C.c_float, C.POINTER(OrcStats)]
    L.orc_occluded_packet.restype = C.c_uint32
    L.orc_occluded_packet.argtypes = [vp, C.c_uint32, vp, vp, C.c_float, C.POINTER(OrcStats)]
    L.orc_camera_packet.argtypes = [C.POINTER(OrcCamera), u32p, C.c_uint32, C.c_uint32, vp, vp, f32p]
    L.orc_trace_block.argtypes = [vp, C.POINTER(OrcCamera)] + [C.c_uint32] * 7 + [C.c_float, vp, C.POINTER(OrcStats)]
    L.orc_render.argtypes = [vp, C.POINTER(OrcCamera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_int, vp,
                             C.POINTER(OrcStats)]
    L.orc_max_threads.restype = C.c_int
    L.orc_render_rect.argtypes = [vp, C.POINTER(OrcCamera)] + [C.c_uint32] * 7 + [C.c_float, C.c_int, vp, C.POINTER(OrcStats)]
    L.orc_scene_set_env_light.argtypes = [vp, C.c_int32, C.c_int32, vp]
    L.orc_scene_set_env_light.restype = None
    L.orc_x_env_sample.argtypes = [vp, C.c_uint32, vp, vp, vp]
    L.orc_x_env_sample.restype = None
    L.orc_gbuffer_block.argtypes = [vp, C.POINTER(OrcCamera)] + [C.c_uint32] * 6 + [C.c_float, vp]
    L.orc_gbuffer_block.restype = None
    L.orc_set_rr_depth.argtypes = [C.c_uint32]
    L.orc_set_rr_depth.restype = None
    L.orc_set_anyhit_accounting.argtypes = [C.c_int]
    L.orc_set_anyhit_accounting.restype = None
    L.orc_libm_sincos.argtypes = [C.c_uint32, vp, vp, vp]
    L.orc_libm_powf22.argtypes = [C.c_uint32, vp, vp]
    _oracle = L
    return L


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def vptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


# ----------------------------------------------------------------------------- scene description
class MeshDesc:
    """A triangle soup + materials, the unit Bvh::build takes (mesh.h:87-104)."""

    def __init__(self, indices, positions, prim_material, materials, normals=None, texcoords=None):
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        self.positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        self.prim_material = np.ascontiguousarray(prim_material, dtype=np.uint32)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.texcoords = None if texcoords is None else np.ascontiguousarray(texcoords, dtype=np.float32).reshape(-1, 2)
        assert len(self.prim_material) == len(self.indices)

    @property
    def prim_count(self):
        return len(self.indices)

    @property
    def vertex_count(self):
        return len(self.positions)


def make_material(diffuse=(0, 0, 0), emissive=(0, 0, 0), reflection=0, alpha_test=0, diffuse_map=-1, bump_map=-1):
    m = np.zeros((), dtype=MATERIAL_DTYPE)
    m["diffuse"] = diffuse
    m["emissive"] = emissive
    m["reflectionType"] = reflection
    m["alphaTest"] = alpha_test
    m["diffuseMap"] = diffuse_map
    m["bumpMap"] = bump_map
    return m


class SceneDesc:
    def __init__(self, meshes, cam_pos, cam_dir, width, height, light=None, textures=(), exposure=1.0, env=None):
        self.meshes = list(meshes)
        self.cam_pos = np.asarray(cam_pos, dtype=np.float32)
        self.cam_dir = np.asarray(cam_dir, dtype=np.float32)
        self.width, self.height = int(width), int(height)
        self.light = light  # (dir[3], intensity[3]) or None
        self.textures = list(textures)  # uint8 arrays (h, w, comp)
        self.exposure = float(exposure)
        self.env = None if env is None else np.ascontiguousarray(env, dtype=np.float32)  # (h, w, 4) float RGBA environment map

    def write_env(self, path):
        """The raw float environment map the reference harness reads in place of an .exr (oracle/ref_glue.cpp)."""
        h, w, _ = self.env.shape
        with open(path, "wb") as f:
            f.write(b"PRTE")
            f.write(struct.pack("<ii", w, h))
            f.write(self.env.astype("<f4").tobytes())

    def write_prts(self, path):
        with open(path, "wb") as f:
            f.write(b"PRTS")
            f.write(struct.pack("<II", 1, len(self.meshes)))
            for m in self.meshes:
                f.write(struct.pack("<5I", m.prim_count, m.vertex_count, len(m.materials),
                                    int(m.normals is not None), int(m.texcoords is not None)))
                f.write(m.indices.tobytes())
                f.write(m.positions.tobytes())
                if m.normals is not None:
                    f.write(m.normals.tobytes())
                if m.texcoords is not None:
                    f.write(m.texcoords.tobytes())
                f.write(m.prim_material.tobytes())
                f.write(m.materials.tobytes())
            f.write(struct.pack("<I", len(self.textures)))
            for t in self.textures:
                h, w, c = t.shape
                f.write(struct.pack("<3i", w, h, c))
                f.write(np.ascontiguousarray(t, dtype=np.uint8).tobytes())
            f.write(struct.pack("<I", int(self.light is not None)))
            ld, li = self.light if self.light is not None else ((0, 0, 0), (0, 0, 0))
            f.write(np.asarray(ld, dtype="<f4").tobytes())
            f.write(np.asarray(li, dtype="<f4").tobytes())
            f.write(self.cam_pos.astype("<f4").tobytes())
            f.write(self.cam_dir.astype("<f4").tobytes())
            f.write(struct.pack("<IIf", self.width, self.height, self.exposure))


class OracleScene:
    """The oracle's scene + camera built from a SceneDesc."""

    def __init__(self, desc, vertex_normals_for=()):
        L = oracle()
        self.L = L
        self.desc = desc
        self.scene = L.orc_scene_create()
        self.bvhs = []
        for t in desc.textures:
            h, w, c = t.shape
            tt = np.ascontiguousarray(t, dtype=np.uint8)
            L.orc_scene_add_texture(self.scene, w, h, c, vptr(tt))
        for m in desc.meshes:
            om = L.orc_mesh_create(m.prim_count, m.vertex_count, len(m.materials), vptr(m.indices), vptr(m.positions),
                                   vptr(m.normals), vptr(m.texcoords), vptr(m.prim_material), vptr(m.materials))
            b = L.orc_bvh_build(om)
            self.bvhs.append(b)
            L.orc_scene_add(self.scene, b)
        if desc.light is not None:
            L.orc_scene_set_directional_light(self.scene, f3(desc.light[0]), f3(desc.light[1]))
        if desc.env is not None:
            h, w, _ = desc.env.shape
            L.orc_scene_set_env_light(self.scene, w, h, vptr(desc.env))
        self.camera = OrcCamera()
        L.orc_camera_create(C.byref(self.camera), f3(desc.cam_pos), f3(desc.cam_dir), desc.width, desc.height)

    def close(self):
        if self.scene:
            self.L.orc_scene_destroy(self.scene)
            self.scene = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gbuffer(self, kind, rect, seed=12345):
        d = self.desc
        x0, y0, x1, y1 = rect
        rgb = np.zeros((d.height, d.width, 3), dtype=np.float32)
        self.L.orc_gbuffer_block(self.scene, C.byref(self.camera), x0, y0, x1, y1, kind, seed, d.exposure, vptr(rgb))
        return rgb[y0:y1 + 1, x0:x1 + 1].copy()

    def env_tables(self):
        h, w, _ = self.desc.env.shape
        self.L.orc_scene_env_vertical.restype = C.POINTER(C.c_float)
        self.L.orc_scene_env_horizontal.restype = C.POINTER(C.c_float)
        self.L.orc_scene_env_vertical.argtypes = [C.c_void_p]
        self.L.orc_scene_env_horizontal.argtypes = [C.c_void_p]
        v = np.ctypeslib.as_array(self.L.orc_scene_env_vertical(self.scene), shape=(h,)).copy()
        hh = np.ctypeslib.as_array(self.L.orc_scene_env_horizontal(self.scene), shape=(h * w,)).copy()
        return v, hh

    def env_sample(self, u):
        u = np.ascontiguousarray(u, dtype=np.float32).reshape(-1, 2)
        d = np.zeros((len(u), 3), dtype=np.float32)
        c = np.zeros((len(u), 3), dtype=np.float32)
        self.L.orc_x_env_sample(self.scene, len(u), vptr(u), vptr(d), vptr(c))
        return d, c

    def nodes(self, i):
        n = self.L.orc_bvh_node_count(self.bvhs[i])
        p = self.L.orc_bvh_nodes(self.bvhs[i])
        return np.frombuffer(C.string_at(p, n * C.sizeof(OrcNode)), dtype=NODE_DTYPE).copy()

    def prim_remap(self, i):
        n = self.L.orc_bvh_prim_count(self.bvhs[i])
        p = self.L.orc_bvh_prim_remap(self.bvhs[i])
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def radius(self):
        return self.L.orc_scene_radius(self.scene)

    def render(self, spp, max_depth=14, seed=12345, threads=0, stats=True):
        d = self.desc
        rgb = np.zeros((d.height, d.width, 3), dtype=np.float32)
        st = OrcStats()
        self.L.orc_render(self.scene, C.byref(self.camera), spp, max_depth, seed, d.exposure, threads, vptr(rgb),
                          C.byref(st) if stats else None)
        return rgb, st.as_dict()

    def render_rect(self, rect, spp, max_depth=14, seed=12345, threads=0, stats=True):
        """Multi-threaded render of an inclusive rectangle; returns (crop, stats)."""
        d = self.desc
        x0, y0, x1, y1 = rect
        rgb = np.zeros((d.height, d.width, 3), dtype=np.float32)
        st = OrcStats()
        self.L.orc_render_rect(self.scene, C.byref(self.camera), x0, y0, x1, y1, spp, max_depth, seed, d.exposure, threads,
                               vptr(rgb), C.byref(st) if stats else None)
        return rgb[y0:y1 + 1, x0:x1 + 1].copy(), st.as_dict()

    def trace_block(self, x0, y0, x1, y1, spp, max_depth=14, seed=12345):
        d = self.desc
        rgb = np.zeros((d.height, d.width, 3), dtype=np.float32)
        st = OrcStats()
        self.L.orc_trace_block(self.scene, C.byref(self.camera), x0, y0, x1, y1, spp, max_depth, seed, d.exposure,
                               vptr(rgb), C.byref(st))
        return rgb[y0:y1 + 1, x0:x1 + 1].copy(), st.as_dict()

    def intersect_single(self, org, dirs, max_t):
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(len(org), dtype=HIT_DTYPE)
        occ = np.zeros(len(org), dtype=np.uint32)
        h = OrcHit()
        for r in range(len(org)):
            self.L.orc_intersect_single(self.scene, fp(org[r]), fp(dirs[r]), max_t, C.byref(h), None)
            out[r] = (h.t, h.i, h.j, h.k, h.primId, h.meshId)
            occ[r] = self.L.orc_occluded_single(self.scene, fp(org[r]), fp(dirs[r]), max_t, None)
        return out, occ

    def intersect_packet(self, org, dirs, max_t):
        org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 8, 3)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 8, 3)
        out = np.zeros((len(org), 8), dtype=HIT_DTYPE)
        occ = np.zeros((len(org), 8), dtype=np.uint32)
        for g in range(len(org)):
            avg = np.zeros(3, dtype=np.float32)
            for l in range(8):
                avg = (avg + dirs[g, l]).astype(np.float32)
            avg = (avg / np.float32(8.0)).astype(np.float32)
            hits = np.zeros(8, dtype=HIT_DTYPE)
            self.L.orc_intersect_packet(self.scene, vptr(org[g]), vptr(dirs[g]), fp(avg), max_t, vptr(hits), None)
            out[g] = hits
            bits = self.L.orc_occluded_packet(self.scene, 0xFF, vptr(org[g]), vptr(dirs[g]), max_t, None)
            occ[g] = [(bits >> l) & 1 for l in range(8)]
        return out.reshape(-1), occ.reshape(-1)


# ----------------------------------------------------------------------------- reference harness
def ref_binary(name):
    """Path of a built reference harness binary, or None (e.g. /root/reference absent and no prebuilt copy)."""
    p = os.path.join(REF_DIR, name)
    return p if os.path.exists(p) and os.access(p, os.X_OK) else None


def run_ref(name, *args):
    exe = ref_binary(name)
    assert exe, f"{name} not built"
    subprocess.check_call([exe] + [str(a) for a in args], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def ref_rays(desc, org, dirs, max_t):
    """single + packet hits / occlusion from the compiled reference for rays (N multiple of 8, one maxT)."""
    org = np.asarray(org, dtype=np.float32).reshape(-1, 3)
    dirs = np.asarray(dirs, dtype=np.float32).reshape(-1, 3)
    n = len(org)
    assert n % 8 == 0
    with tempfile.TemporaryDirectory() as td:
        sp, ip, op = (os.path.join(td, x) for x in ("s.prts", "in.bin", "out.bin"))
        desc.write_prts(sp)
        rec = np.concatenate([org, dirs, np.full((n, 1), max_t, dtype=np.float32)], axis=1).astype("<f4")
        rec.tofile(ip)
        run_ref("ref_core", "rays", sp, ip, op)
        raw = np.fromfile(op, dtype="<u4").reshape(n, 16)
    single = np.zeros(n, dtype=HIT_DTYPE)
    packet = np.zeros(n, dtype=HIT_DTYPE)
    for k, (name, _) in enumerate(HIT_DTYPE.descr):
        single[name] = raw[:, k].view(HIT_DTYPE[name])
        packet[name] = raw[:, 7 + k].view(HIT_DTYPE[name])
    return single, raw[:, 6].copy(), packet, raw[:, 13].copy()


def ref_bvh(desc, threaded=False):
    with tempfile.TemporaryDirectory() as td:
        sp, op = os.path.join(td, "s.prts"), os.path.join(td, "out.bin")
        desc.write_prts(sp)
        run_ref("ref_core", "bvh", sp, op, *(["threaded"] if threaded else []))
        buf = open(op, "rb").read()
    off = 0
    (mc,) = struct.unpack_from("<I", buf, off)
    off += 4
    out = []
    for _ in range(mc):
        nc, lc, pc = struct.unpack_from("<3I", buf, off)
        off += 12
        nodes = np.frombuffer(buf, dtype=NODE_DTYPE, count=nc, offset=off).copy()
        off += nc * NODE_DTYPE.itemsize
        remap = np.frombuffer(buf, dtype="<u4", count=pc, offset=off).copy()
        off += pc * 4
        bbox = np.frombuffer(buf, dtype="<f4", count=6, offset=off).copy()
        off += 24
        out.append(dict(nodes=nodes, leaf_count=lc, remap=remap, bbox=bbox))
    sbox = np.frombuffer(buf, dtype="<f4", count=6, offset=off).copy()
    (radius,) = struct.unpack_from("<f", buf, off + 24)
    return out, sbox, radius


def ref_gbuffer(desc, kind, rect, seed=12345):
    """GbufferVisualizer::TraceBlock of the compiled reference (kind 0 diffuse, 1 mesh normal, 2 normal), per-pixel states."""
    x0, y0, x1, y1 = rect
    with tempfile.TemporaryDirectory() as td:
        sp, op = os.path.join(td, "s.prts"), os.path.join(td, "out.bin")
        desc.write_prts(sp)
        run_ref("ref_path", "gbuffer", sp, kind, x0, y0, x1, y1, seed, op)
        return np.fromfile(op, dtype="<f4").reshape(y1 - y0 + 1, x1 - x0 + 1, 3).copy()


def ref_envlight(desc, u):
    """InfiniteAreaLight::create + sample of the compiled reference: (verticalP, horizontalP, dir, color)."""
    u = np.ascontiguousarray(u, dtype=np.float32).reshape(-1, 2)
    with tempfile.TemporaryDirectory() as td:
        ep, up, op = os.path.join(td, "e.prte"), os.path.join(td, "u.bin"), os.path.join(td, "out.bin")
        desc.write_env(ep)
        u.astype("<f4").tofile(up)
        run_ref("ref_path", "envlight", ep, up, op)
        raw = np.fromfile(op, dtype="<f4")
    w, h = (int(v) for v in raw[:2].view(np.int32))
    vp = raw[2:2 + h].copy()
    hp = raw[2 + h:2 + h + w * h].copy()
    rec = raw[2 + h + w * h:].reshape(-1, 6)
    return vp, hp, rec[:, :3].copy(), rec[:, 3:].copy()


def ref_render(desc, spp, rect, seed=12345, threads=0, stats=True):
    x0, y0, x1, y1 = rect
    with tempfile.TemporaryDirectory() as td:
        sp, op = os.path.join(td, "s.prts"), os.path.join(td, "out.bin")
        desc.write_prts(sp)
        extra = []
        if desc.env is not None:
            ep = os.path.join(td, "e.prte")
            desc.write_env(ep)
            extra = [ep]
        run_ref("ref_path_stats" if stats else "ref_path", "render", sp, spp, x0, y0, x1, y1, seed, threads, op, *extra)
        raw = open(op, "rb").read()
    n = (x1 - x0 + 1) * (y1 - y0 + 1) * 3
    rgb = np.frombuffer(raw, dtype="<f4", count=n).reshape(y1 - y0 + 1, x1 - x0 + 1, 3).copy()
    rays, occl = struct.unpack_from("<QQ", raw, n * 4)
    (sec,) = struct.unpack_from("<d", raw, n * 4 + 16)
    return rgb, dict(raysTraced=rays, occludedTraced=occl, seconds=sec)


# ----------------------------------------------------------------------------- fixtures
def load_cornell_mesh():
    """Cornell box as SampleModels::getCornellBox(true) builds it (sample_models.cpp:11-207); data dumped from the
    compiled reference by tests/golden/make_golden.py."""
    z = np.load(os.path.join(GOLDEN, "cornell_box.npz"))
    return MeshDesc(z["indices"], z["positions"], z["prim_material"], z["materials"].view(MATERIAL_DTYPE).reshape(-1))


def load_teapot_mesh(scale=0.005, translate=(-0.5, 0.0, 0.5)):
    """The reference's data/teapot/teapot.obj, fan-triangulated, transformed as main.cpp:40-44, specular material
    kd 0.9 (main.cpp:31-35).  Vertex normals must still be computed (calculateVertexNormals)."""
    z = np.load(os.path.join(GOLDEN, "teapot_mesh.npz"))
    pos = z["positions"].astype(np.float32)
    s = np.float32(scale)
    pos = (s * pos + np.asarray(translate, dtype=np.float32)).astype(np.float32)
    mats = np.array([make_material(diffuse=(0.9, 0.9, 0.9), reflection=1)], dtype=MATERIAL_DTYPE)
    idx = z["indices"]
    return MeshDesc(idx, pos, np.zeros(len(idx), dtype=np.uint32), mats, texcoords=z["texcoords"])


def oracle_vertex_normals(mesh):
    """calculateVertexNormals (mesh.cpp:108-149) through the oracle; returns a new MeshDesc with normals."""
    L = oracle()
    om = L.orc_mesh_create(mesh.prim_count, mesh.vertex_count, len(mesh.materials), vptr(mesh.indices),
                           vptr(mesh.positions), None, vptr(mesh.texcoords), vptr(mesh.prim_material), vptr(mesh.materials))
    L.orc_mesh_calculate_vertex_normals(om)
    n = np.ctypeslib.as_array(L.orc_mesh_normals(om), shape=(mesh.vertex_count, 3)).copy()
    L.orc_mesh_destroy(om)
    return MeshDesc(mesh.indices, mesh.positions, mesh.prim_material, mesh.materials, normals=n, texcoords=mesh.texcoords)


def cornell_scene(width, height, with_teapot=True):
    """setupCornellBox (main.cpp:22-55)."""
    meshes = [load_cornell_mesh()]
    if with_teapot:
        meshes.append(oracle_vertex_normals(load_teapot_mesh()))
    return SceneDesc(meshes, cam_pos=(0, 0.965, 2.6), cam_dir=(0, 0, -1.0), width=width, height=height)


def sky_env(w, h, seed=7, black_rows=False):
    """Seeded procedural float RGBA environment map (h, w, 4): gradient sky, a bright sun blob, per-texel noise.
    black_rows adds all-black rows and columns (zero-pdf entries, NaN rows in the horizontal CDF, light.cpp:63-71)."""
    rng = np.random.default_rng(seed)
    y = (np.arange(h) + 0.5) / h
    x = (np.arange(w) + 0.5) / w
    e = np.zeros((h, w, 4), dtype=np.float32)
    e[..., 0] = 0.3 + 0.5 * (1 - y)[:, None]
    e[..., 1] = 0.4 + 0.4 * (1 - y)[:, None]
    e[..., 2] = 0.9 * (1 - 0.5 * y)[:, None]
    sun = np.exp(-(((x[None, :] - 0.3) * 6) ** 2 + ((y[:, None] - 0.25) * 6) ** 2))
    e[..., :3] += (40.0 * sun)[..., None].astype(np.float32)
    e[..., :3] *= rng.uniform(0.5, 1.5, size=(h, w, 1)).astype(np.float32)
    if black_rows:
        e[h // 2:, :, :3] = 0.0
        e[1, :, :3] = 0.0
        e[:, : w // 8, :3] = 0.0
    e[..., 3] = 1.0
    return e


def env_test_u(n, seed=11):
    """(u.x, u.y) pairs for InfiniteAreaLight::sample incl. the ends of [0, 1)."""
    rng = np.random.default_rng(seed)
    u = rng.random((n, 2)).astype(np.float32)
    top = np.float32(0.99999994)
    u[:64] = top
    u[64:128, 0] = 0.0
    u[128:192, 1] = 0.0
    u[192:224] = 0.0
    u[224:256, 0] = top
    u[256:288, 1] = top
    return u


def scene_desc_from_product(scene, camera, exposure=1.0):
    """SceneDesc (for the oracle / reference harness) holding the same arrays the product uploads."""
    a = scene.arrays()
    meshes = [MeshDesc(m["indices"], m["positions"], m["prim_material"], m["materials"].view(MATERIAL_DTYPE), normals=m["normals"],
                       texcoords=m["texcoords"]) for m in a["meshes"]]
    light = (a["light_dir"], a["light_intensity"]) if a["has_light"] else None
    # Camera::create's ARGUMENTS (camera.desc.dir is already normalised; the basis depends on the raw direction)
    d = SceneDesc(meshes, camera.pos_arg, camera.dir_arg, camera.width, camera.height, light=light, textures=a["textures"],
                  exposure=exposure, env=a["env"])
    d.product_arrays = a
    return d


def teapot_product_mesh():
    import prt_amd
    m = load_teapot_mesh(scale=1.0, translate=(0, 0, 0))
    pm = prt_amd.Mesh.from_arrays(m.indices, m.positions, m.prim_material, m.materials.view(prt_amd.MATERIAL_DTYPE),
                                  texcoords=m.texcoords)
    pm.transform(0.005, (-0.5, 0.0, 0.5))
    return pm


def build_fake_rccl():
    """tests/fake_rccl.cpp -> tests/_build/libfake_rccl.so (host C++ against libamdhip64; rebuilt when the source is newer)."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src, out = os.path.join(here, "fake_rccl.cpp"), os.path.join(here, "_build", "libfake_rccl.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", src, "-o", out + ".tmp",
                               f"-L{rocm}/lib", "-lamdhip64", "-pthread", f"-Wl,-rpath,{rocm}/lib"])
        os.replace(out + ".tmp", out)
    return out
