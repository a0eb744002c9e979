"""Shared test plumbing: ctypes bindings for the oracle (checker only), a small independent
OBJ/MTL/JSON reader that follows /root/reference/src/scene/loader.cpp:144-294, and synthetic
scene generators.  Nothing here is imported by the product path."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(GOLDEN, "scenes")
ORACLE_DIR = os.path.join(ROOT, "oracle")

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


def cornell_moved(glass=True):
    """The Cornell box with the two blocks and the light moved by object transforms: rotation about
    all three axes, a non-uniform scale, a translation (SceneObject::translate/rotate/scale,
    src/scene/core.hpp:177-240; the shipped scene files leave them at identity)."""
    objs, cam = cornell(glass=glass)
    by_name = {o["name"]: o for o in objs}
    names = list(by_name)
    tall = [n for n in names if "tall" in n.lower()][0]
    short = [n for n in names if "short" in n.lower()][0]
    light = [n for n in names if "light" in n.lower()][0]
    by_name[tall]["trs"] = ([0.12, 0.0, -0.05], [0.0, 17.0, 4.0], [0.9, 1.15, 1.0])
    by_name[short]["trs"] = ([-0.1, 0.05, 0.1], [8.0, -31.0, 0.0], [1.2, 0.8, 1.1])
    by_name[light]["trs"] = ([0.05, -0.01, 0.0], [0.0, 30.0, 0.0], [0.8, 1.0, 0.8])
    return objs, cam


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


def write_ppm(path, rgba):
    """(h, w) uint32 RGBA8 -> binary PPM (alpha dropped)."""
    h, w = rgba.shape
    rgb = np.stack([(rgba >> s) & 255 for s in (0, 8, 16)], axis=-1).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(rgb.tobytes())


def textured_box_files(directory, bsdf_names=("Kd", "Ks", "Ke")):
    """A closed room with mapped materials, written as .obj/.mtl/.ppm: floor with a diffuse map,
    back wall with diffuse + specular maps, ceiling lamp with an emission map (and a constant Ke,
    which is what makes it a light), an untextured block.  Returns (obj path, camera dict, maps)
    where maps[material name] = {"map_Kd": array, ...}."""
    rng = np.random.Generator(np.random.PCG64(9))
    def image(w, h, lo, hi):
        c = rng.integers(lo, hi, (h, w, 3), dtype=np.uint32)
        yy, xx = np.mgrid[0:h, 0:w]
        c[((xx // 3 + yy // 2) % 2) == 0] //= 3
        return (c[..., 0] | (c[..., 1] << 8) | (c[..., 2] << 16) | (255 << 24)).astype(np.uint32)
    maps = {"floor": {"map_Kd": image(16, 12, 40, 256)},
            "wall": {"map_Kd": image(7, 5, 60, 230), "map_Ks": image(4, 4, 0, 90)},
            "lamp": {"map_Ke": image(8, 8, 120, 256)}}
    mtl = []
    for name, consts in (("floor", "Kd 0.3 0.3 0.3\nKs 0 0 0"), ("wall", "Kd 0.5 0.5 0.5\nKs 0.2 0.2 0.2\nNs 20"),
                         ("lamp", "Kd 0 0 0\nKs 0 0 0\nKe 9 9 9"), ("plain", "Kd 0.6 0.3 0.2\nKs 0.1 0.1 0.1\nNs 8")):
        mtl.append("newmtl %s\n%s" % (name, consts))
        for key, img in maps.get(name, {}).items():
            fname = "%s_%s.ppm" % (name, key)
            write_ppm(os.path.join(directory, fname), img)
            mtl.append("%s %s" % (key, fname))
        mtl.append("")
    open(os.path.join(directory, "room.mtl"), "w").write("\n".join(mtl))
    def quad(name, mat, p, n, uvscale=1.0):
        return ["o %s" % name, "usemtl %s" % mat] + ["v %g %g %g" % tuple(q) for q in p] + \
               ["vt %g %g" % (u * uvscale, v * uvscale) for u, v in ((0, 0), (1, 0), (1, 1), (0, 1))] + ["vn %g %g %g" % tuple(n)] * 4 + \
               ["f -4/-4/-4 -3/-3/-3 -2/-2/-2", "f -4/-4/-4 -2/-2/-2 -1/-1/-1"]
    lines = ["mtllib room.mtl"]
    lines += quad("floor", "floor", [(-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1)], (0, 1, 0), 2.5)      # uv beyond [0,1]: wrap
    lines += quad("back", "wall", [(-1, 0, 1), (1, 0, 1), (1, 2, 1), (-1, 2, 1)], (0, 0, -1))
    lines += quad("left", "plain", [(-1, 0, -1), (-1, 0, 1), (-1, 2, 1), (-1, 2, -1)], (1, 0, 0))
    lines += quad("right", "plain", [(1, 0, 1), (1, 0, -1), (1, 2, -1), (1, 2, 1)], (-1, 0, 0))
    lines += quad("ceiling", "plain", [(-1, 2, 1), (1, 2, 1), (1, 2, -1), (-1, 2, -1)], (0, -1, 0))
    lines += quad("lamp", "lamp", [(-0.4, 1.98, 0.4), (0.4, 1.98, 0.4), (0.4, 1.98, -0.4), (-0.4, 1.98, -0.4)], (0, -1, 0))
    lines += quad("block", "plain", [(-0.5, 0.4, 0.2), (0.3, 0.4, 0.0), (0.4, 0.4, 0.6), (-0.3, 0.4, 0.7)], (0, 1, 0))
    obj = os.path.join(directory, "room.obj")
    open(obj, "w").write("\n".join(lines) + "\n")
    cam = dict(pos=np.array([0.0, 1.0, -2.6], np.float32), look_at=np.array([0.0, 0.9, 0.0], np.float32),
               up=np.array([0, 1, 0], np.float32), fov=45.0, width=64, height=64)
    return obj, cam, maps


# --------------------------------------------------------------------------- synthetic scenes
def pcg_floats(n, seed):
    """n uniform floats in [0,1) from a 64-bit LCG/xorshift stream (numpy only, deterministic)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random(n, dtype=np.float32)


def random_soup(ntris, seed=12345, half_extent=0.0025):
    """Small triangles with centres uniform in [0,1]^3; every triangle has its own 3 vertices."""
    r = pcg_floats(ntris * 12, seed).reshape(ntris, 12)
    centre = r[:, 0:3]
    off = (r[:, 3:12].reshape(ntris, 3, 3) - np.float32(0.5)) * np.float32(2 * half_extent)
    pos = (centre[:, None, :] + off).astype(np.float32).reshape(-1, 3)
    e1 = pos[1::3] - pos[0::3]
    e2 = pos[2::3] - pos[0::3]
    nrm = np.cross(e1, e2)
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    ln[ln == 0] = 1
    nrm = np.repeat((nrm / ln).astype(np.float32), 3, axis=0)
    uv = np.tile(np.array([[0, 0], [1, 0], [0, 1]], dtype=np.float32), (ntris, 1))
    idx = np.arange(3 * ntris, dtype=np.uint32)
    return dict(name="soup", positions=np.ascontiguousarray(pos), normals=np.ascontiguousarray(nrm),
                texcoords=np.ascontiguousarray(uv), indices=idx)


def heightfield(nx, nz, seed=7, amplitude=0.08):
    """Tessellated grid of 2*nx*nz triangles over [0,1]^2 with shared vertices, smooth bumps in y."""
    xs = np.linspace(0, 1, nx + 1, dtype=np.float32)
    zs = np.linspace(0, 1, nz + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, zs, indexing="xy")
    Y = (amplitude * (np.sin(X * 9.0 + seed) * np.cos(Z * 7.0) + 0.5 * np.sin(X * 23.0) * np.sin(Z * 19.0))).astype(np.float32)
    pos = np.stack([X, Y, Z], axis=-1).reshape(-1, 3).astype(np.float32)
    dx = np.gradient(Y, axis=1) * nx
    dz = np.gradient(Y, axis=0) * nz
    nrm = np.stack([-dx, np.ones_like(Y), -dz], axis=-1).reshape(-1, 3)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    uv = np.stack([X, Z], axis=-1).reshape(-1, 2).astype(np.float32)
    i = np.arange(nx, dtype=np.uint32)[None, :] + (nx + 1) * np.arange(nz, dtype=np.uint32)[:, None]
    a, b, c, d = i, i + 1, i + nx + 1, i + nx + 2
    idx = np.stack([a, c, b, b, c, d], axis=-1).reshape(-1).astype(np.uint32)
    return dict(name="heightfield", positions=np.ascontiguousarray(pos), normals=np.ascontiguousarray(nrm),
                texcoords=np.ascontiguousarray(uv), indices=np.ascontiguousarray(idx))


def quad_light(y=1.5, half=0.25, centre=(0.5, 0.5), emission=(17.0, 12.0, 4.0)):
    cx, cz = centre
    pos = np.array([[cx - half, y, cz - half], [cx + half, y, cz - half], [cx + half, y, cz + half], [cx - half, y, cz + half]],
                   dtype=np.float32)
    nrm = np.tile(np.array([[0, -1, 0]], dtype=np.float32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    mat = np.array(list(emission) + [0, 0, 0] + [0, 0, 0] + [0, 0, 0] + [1.0, 1.5, 1.0, 2], dtype=np.float32)
    return dict(name="light", positions=pos, normals=nrm, texcoords=uv, indices=idx, material=mat)


def diffuse_material(kd=(0.7, 0.7, 0.7), bsdf_type=2):
    return np.array([0, 0, 0] + list(kd) + [0, 0, 0] + [0, 0, 0] + [1.0, 1.5, 1.0, bsdf_type], dtype=np.float32)


def synthetic_scene(kind, ntris, seed=12345):
    """[big mesh, small emissive quad] + a camera looking at the unit cube."""
    if kind == "soup":
        mesh = random_soup(ntris, seed)
    else:
        nx = int(round((ntris / 2) ** 0.5))
        mesh = heightfield(nx, max(1, ntris // (2 * nx)), seed)
    mesh["material"] = diffuse_material()
    cam = dict(pos=np.array([0.5, 0.8, -0.5], dtype=np.float32), look_at=np.array([0.5, 0.0, 0.5], dtype=np.float32),
               up=np.array([0, 1, 0], dtype=np.float32), fov=55.0, width=256, height=256)
    return [mesh, quad_light()], cam


def boxes_of_triangles(pos, idx):
    """Leaf AABBs (n,8) with w = 1 as src/scene/mesh.cpp:88-98 fills them, and the merged box."""
    p = pos[idx.reshape(-1, 3)]
    lo, hi = p.min(axis=1), p.max(axis=1)
    n = len(lo)
    boxes = np.ones((n, 8), dtype=np.float32)
    boxes[:, 0:3] = lo
    boxes[:, 4:7] = hi
    merged = np.ones(8, dtype=np.float32)
    merged[0:3] = pos.min(axis=0)
    merged[4:7] = pos.max(axis=0)
    return boxes, merged
