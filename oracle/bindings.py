"""TEST INFRASTRUCTURE, not product code: ctypes bindings of the CPU oracle (liboracle.so, and libref.so where it was
built) and a small independent OBJ/MTL/JSON reader that follows /root/reference/src/scene/loader.cpp:144-294, so that
the oracle can be fed the scenes the product loads.  Imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py only; nothing in cuda-path-tracer_amd/ imports it."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(GOLDEN, "scenes")
ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))

f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")

IDENTITY = np.eye(4, dtype=np.float32).reshape(-1).copy()


def _build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True, capture_output=True)


_oracle = None


def oracle():
    """liboracle.so -- the CPU restatement (oracle/oracle.cpp)."""
    global _oracle
    if _oracle is not None:
        return _oracle
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        _build_oracle()
    L = C.CDLL(path)
    L.orc_morton_key.restype = C.c_uint64
    L.orc_morton_key.argtypes = [f32p, f32p, C.c_uint32]
    L.orc_build_accel.argtypes = [u32p, f32p, f32p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.orc_scene_create.restype = C.c_void_p
    L.orc_scene_destroy.argtypes = [C.c_void_p]
    L.orc_scene_add_mesh.argtypes = [C.c_void_p, f32p, f32p, f32p, C.c_uint32, u32p, C.c_uint32]
    L.orc_scene_add_instance.argtypes = [C.c_void_p, C.c_int, f32p, f32p]
    L.orc_scene_set_camera.argtypes = [C.c_void_p, f32p, f32p, f32p, C.c_float]
    L.orc_scene_build.argtypes = [C.c_void_p]
    L.orc_scene_get_camera.argtypes = [C.c_void_p, f32p]
    L.orc_scene_num_lights.restype = C.c_uint32
    L.orc_scene_num_lights.argtypes = [C.c_void_p]
    L.orc_scene_get_accel.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_first_hit.argtypes = [C.c_void_p, f32p, C.c_uint64, u32p, u32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_occlude.argtypes = [C.c_void_p, f32p, C.c_uint64, u8p]
    L.orc_camera_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, f32p]
    L.orc_render.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                             C.c_int, C.c_void_p]
    L.orc_tea.restype = C.c_uint32
    L.orc_tea.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_next1d_seq.argtypes = [C.c_uint32, C.c_uint32, f32p]
    L.orc_bbox_intersect.restype = C.c_int
    L.orc_bbox_intersect.argtypes = [f32p, f32p, f32p]
    L.orc_bsdf_sample.argtypes = [C.c_int, f32p, f32p, C.c_float, C.c_float, C.c_float, f32p]
    L.orc_bsdf_eval.argtypes = [C.c_int, f32p, f32p, f32p, f32p]
    L.orc_material_bsdf.argtypes = [f32p, f32p]
    L.orc_mesh_vertex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_float, C.c_float, f32p]
    L.orc_frame.argtypes = [f32p, f32p]
    L.orc_trs_matrix.argtypes = [f32p, f32p, f32p, f32p, f32p]
    u32arr = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
    L.orc_scene_add_texture.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u32arr]
    L.orc_scene_set_maps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_tex2d.argtypes = [C.c_uint32, C.c_uint32, u32arr, C.c_float, C.c_float, f32p]
    L.orc_luminance.restype = C.c_float
    L.orc_luminance.argtypes = [C.c_float, C.c_float, C.c_float]
    _oracle = L
    return L


_ref = None


def ref_lib():
    """oracle/_ref/libref.so -- the reference's own headers compiled for the host, or None."""
    global _ref
    if _ref is not None:
        return _ref
    path = os.path.join(ORACLE_DIR, "_ref", "libref.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.ref_sizes.argtypes = [u32p]
    L.ref_scene_create.restype = C.c_void_p
    L.ref_scene_destroy.argtypes = [C.c_void_p]
    L.ref_scene_add_mesh.argtypes = [C.c_void_p, f32p, f32p, f32p, C.c_uint32, u32p, C.c_uint32, u32p, f32p]
    L.ref_scene_add_instance.argtypes = [C.c_void_p, C.c_int, f32p, f32p]
    L.ref_scene_set_top.argtypes = [C.c_void_p, u32p, f32p]
    L.ref_first_hit.argtypes = [C.c_void_p, f32p, C.c_uint64, u32p, u32p, C.c_void_p, C.c_void_p]
    L.ref_occlude.argtypes = [C.c_void_p, f32p, C.c_uint64, u8p]
    L.ref_bbox_intersect.restype = C.c_int
    L.ref_bbox_intersect.argtypes = [f32p, f32p, f32p]
    L.ref_mesh_vertex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_float, C.c_float, f32p]
    L.ref_tea.restype = C.c_uint32
    L.ref_tea.argtypes = [C.c_uint32, C.c_uint32]
    L.ref_next1d_seq.argtypes = [C.c_uint32, C.c_uint32, f32p]
    L.ref_bsdf_sample.argtypes = [C.c_int, f32p, f32p, C.c_float, C.c_float, C.c_float, f32p]
    L.ref_bsdf_eval.argtypes = [C.c_int, f32p, f32p, f32p, f32p]
    L.ref_camera_ray.argtypes = [f32p, C.c_float, C.c_float, C.c_float, f32p]
    L.ref_frame.argtypes = [f32p, f32p]
    L.ref_trs_matrix.argtypes = [f32p, f32p, f32p, f32p, f32p]
    L.ref_luminance.restype = C.c_float
    L.ref_luminance.argtypes = [C.c_float, C.c_float, C.c_float]
    _ref = L
    return L


# --------------------------------------------------------------------------- scene files
def _parse_mtl(path):
    """MTL entries in file order with tinyobjloader's defaults (ior 1, shininess 1, Tf 0;
    thirdparty/tinyobjloader/tiny_obj_loader.h:1366-1368)."""
    mats = []
    cur = None
    if not os.path.exists(path):
        return mats
    for line in open(path):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "newmtl":
            cur = dict(name=" ".join(t[1:]), Kd=[0.0] * 3, Ks=[0.0] * 3, Tf=[0.0] * 3, Ke=[0.0] * 3, Ni=1.0, Ns=1.0)
            mats.append(cur)
        elif cur is None:
            continue
        elif t[0] in ("Kd", "Ks", "Ke"):
            cur[t[0]] = [float(x) for x in t[1:4]]
        elif t[0] in ("Kt", "Tf"):
            cur["Tf"] = [float(x) for x in t[1:4]]
        elif t[0] == "Ni":
            cur["Ni"] = float(t[1])
        elif t[0] == "Ns":
            cur["Ns"] = float(t[1])
    return mats


def material_block(m, bsdf_type):
    """16 floats: emission diffuse specular transmittance shininess ior opacity bsdf_type
    (loader.cpp:231-250; opacity rule at :244)."""
    f = np.float32
    tf = [f(x) for x in m["Tf"]]
    opacity = 1.0 if (f(m["Ni"]) == f(1.0) or all(x == 0 for x in tf)) else 0.0
    return np.array(list(m["Ke"]) + list(m["Kd"]) + list(m["Ks"]) + list(m["Tf"]) + [m["Ns"], m["Ni"], opacity, bsdf_type],
                    dtype=np.float32)


DEFAULT_MATERIAL = dict(name="default", Ke=[0.0] * 3, Kd=[0.5] * 3, Ks=[0.5] * 3, Tf=[0.0] * 3, Ns=1.0, Ni=1.5)


def load_obj_scene(path, bsdf_type=2):
    """Objects in first-appearance order, one per (shape, material), vertices de-duplicated on
    (v, vn, vt) in first-use order (loader.cpp:171-211).  Triangles and quads only."""
    base = os.path.dirname(path)
    V, VN, VT = [], [], []
    mats, mat_index = [], {}
    shapes = []          # [name, [(idx triple x3, material_id)]]
    cur_shape = None
    cur_mat = -1

    def shape():
        nonlocal cur_shape
        if cur_shape is None:
            cur_shape = ["", []]
            shapes.append(cur_shape)
        return cur_shape

    for line in open(path):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "v":
            V.append([float(x) for x in t[1:4]])
        elif t[0] == "vn":
            VN.append([float(x) for x in t[1:4]])
        elif t[0] == "vt":
            VT.append([float(x) for x in t[1:3]])
        elif t[0] == "mtllib":
            mats = _parse_mtl(os.path.join(base, t[1]))
            mat_index = {m["name"]: i for i, m in enumerate(mats)}
        elif t[0] == "usemtl":
            cur_mat = mat_index.get(" ".join(t[1:]), -1)
        elif t[0] in ("o", "g"):
            cur_shape = [" ".join(t[1:]), []]
            shapes.append(cur_shape)
        elif t[0] == "f":
            corners = []
            for c in t[1:]:
                p = (c.split("/") + ["", ""])[:3]
                vi = int(p[0])
                ti = int(p[1]) if p[1] else 0
                ni = int(p[2]) if p[2] else 0
                vi = vi - 1 if vi > 0 else len(V) + vi
                ti = (ti - 1 if ti > 0 else len(VT) + ti) if p[1] else -1
                ni = (ni - 1 if ni > 0 else len(VN) + ni) if p[2] else -1
                corners.append((vi, ni, ti))
            if len(corners) == 3:
                tris = [corners]
            elif len(corners) == 4:
                v = [np.array(V[c[0]], dtype=np.float32) for c in corners]
                e02, e13 = v[2] - v[0], v[3] - v[1]
                s02 = np.float32(e02[0] * e02[0] + e02[1] * e02[1] + e02[2] * e02[2])
                s13 = np.float32(e13[0] * e13[0] + e13[1] * e13[1] + e13[2] * e13[2])
                c = corners
                tris = [[c[0], c[1], c[2]], [c[0], c[2], c[3]]] if s02 < s13 else [[c[0], c[1], c[3]], [c[1], c[2], c[3]]]
            else:
                raise ValueError("polygons with more than 4 corners are not handled by the test reader")
            for tri in tris:
                shape()[1].append((tri, cur_mat))

    has_normal, has_uv = len(VN) > 0, len(VT) > 0
    objects = []
    for name, faces in shapes:
        if not faces:
            continue
        per_mat = {}
        for tri, mid in faces:
            if mid not in per_mat:
                mname = "default" if mid < 0 else mats[mid]["name"]
                per_mat[mid] = dict(name=name + "-" + mname, material_id=mid, pos=[], nrm=[], uv=[], idx=[], map={})
                objects.append(per_mat[mid])
            o = per_mat[mid]
            for key in tri:
                if key not in o["map"]:
                    o["map"][key] = len(o["pos"])
                    o["pos"].append(V[key[0]])
                    o["nrm"].append(VN[key[1]] if has_normal else [0.0, 0.0, 0.0])
                    o["uv"].append(VT[key[2]] if has_uv else [0.0, 0.0])
                o["idx"].append(o["map"][key])
    out = []
    for o in objects:
        m = DEFAULT_MATERIAL if o["material_id"] < 0 else mats[o["material_id"]]
        bt = 2 if o["material_id"] < 0 else bsdf_type     # default Material keeps bsdf_type 2 (material.hpp:32)
        out.append(dict(name=o["name"],
                        positions=np.ascontiguousarray(o["pos"], dtype=np.float32).reshape(-1, 3),
                        normals=np.ascontiguousarray(o["nrm"], dtype=np.float32).reshape(-1, 3),
                        texcoords=np.ascontiguousarray(o["uv"], dtype=np.float32).reshape(-1, 2),
                        indices=np.ascontiguousarray(o["idx"], dtype=np.uint32),
                        material=material_block(m, bt),
                        material_name=m["name"]))
    return out


def load_json_camera(path):
    cam = json.load(open(path))["camera"]
    res = cam.get("resolution", [1280, 720])
    return dict(pos=np.array(cam["pos"], dtype=np.float32), look_at=np.array(cam["look_at"], dtype=np.float32),
                up=np.array(cam["up"], dtype=np.float32), fov=float(cam["fov"]), width=int(res[0]), height=int(res[1]))


# glm::inverse of the identity has negative zeros off the diagonal; objects without a transform get that matrix too
IDENTITY_TRS = ([0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])


def trs_matrix(lib, prefix, trs):
    """(transform16, inverse16), column major, from (translate, rotate in degrees, scale)."""
    t, r, s = (np.ascontiguousarray(x, dtype=np.float32) for x in trs)
    m = np.zeros(16, np.float32); inv = np.zeros(16, np.float32)
    getattr(lib, prefix + "_trs_matrix")(t, r, s, m, inv)
    return m, inv


def object_transform(o):
    """An object's transform and inverse by the oracle's restatement of SceneObject::GetTransform."""
    return trs_matrix(oracle(), "orc", o.get("trs", IDENTITY_TRS))


class OracleScene:
    """A scene inside the oracle, fed with flat arrays."""

    def __init__(self, objects, camera=None):
        self.L = oracle()
        self.h = self.L.orc_scene_create()
        self.objects = objects
        for o in objects:
            m = self.L.orc_scene_add_mesh(self.h, o["positions"], o["normals"], o["texcoords"], len(o["positions"]),
                                          o["indices"], len(o["indices"]) // 3)
            inst = self.L.orc_scene_add_instance(self.h, m, object_transform(o)[0], o["material"])
            numbers = []
            for key in ("map_Kd", "map_Ks", "map_Ke"):        # (h, w) uint32 RGBA8 arrays
                img = o.get(key)
                numbers.append(0 if img is None else self.L.orc_scene_add_texture(self.h, img.shape[1], img.shape[0], np.ascontiguousarray(img)))
            if any(numbers):
                self.L.orc_scene_set_maps(self.h, inst, *numbers)
        if camera is not None:
            self.L.orc_scene_set_camera(self.h, camera["pos"], camera["look_at"], camera["up"], camera["fov"])
        self.L.orc_scene_build(self.h)

    def __del__(self):
        try:
            self.L.orc_scene_destroy(self.h)
        except Exception:
            pass

    def accel(self, mesh):
        n = len(self.objects) if mesh < 0 else len(self.objects[mesh]["indices"]) // 3
        nodes = np.zeros((2 * n - 1, 2), dtype=np.uint32)
        boxes = np.zeros((2 * n - 1, 8), dtype=np.float32)
        keys = np.zeros(n, dtype=np.uint64)
        self.L.orc_scene_get_accel(self.h, mesh, nodes.ctypes.data, boxes.ctypes.data, keys.ctypes.data)
        return nodes, boxes, keys

    def camera_rays(self, W, H, spp=1):
        rays = np.zeros((W * H, 8), dtype=np.float32)
        self.L.orc_camera_rays(self.h, W, H, spp, rays)
        return rays

    def first_hit(self, rays, nthreads=8):
        n = len(rays)
        inst = np.zeros(n, dtype=np.uint32)
        prim = np.zeros(n, dtype=np.uint32)
        t = np.zeros(n, dtype=np.float32)
        uv = np.zeros((n, 2), dtype=np.float32)
        cnt = np.zeros(4, dtype=np.uint64)
        self.L.orc_first_hit(self.h, np.ascontiguousarray(rays, dtype=np.float32), n, inst, prim, t.ctypes.data, uv.ctypes.data,
                             cnt.ctypes.data, nthreads)
        return inst, prim, t, uv, cnt

    def occlude(self, rays):
        out = np.zeros(len(rays), dtype=np.uint8)
        self.L.orc_occlude(self.h, np.ascontiguousarray(rays, dtype=np.float32), len(rays), out)
        return out

    def render(self, W, H, spp_begin, spp_end, max_depth, channel=0, sum_mode=0, nthreads=8, output=None):
        out = np.zeros((H, W, 4), dtype=np.float32) if output is None else output
        cnt = np.zeros(4, dtype=np.uint64)
        self.L.orc_render(self.h, out.reshape(-1), W, H, spp_begin, spp_end, max_depth, channel, sum_mode, nthreads, cnt.ctypes.data)
        return out, cnt


def cornell(glass=False, bsdf_type=2):
    d = os.path.join(SCENES, "cornell-box-glass" if glass else "cornell-box")
    return load_obj_scene(os.path.join(d, "cornell-box.obj"), bsdf_type), load_json_camera(os.path.join(d, "cornell-box.json"))


