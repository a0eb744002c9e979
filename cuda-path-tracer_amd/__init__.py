"""ctypes binding of libcpt_amd.so -- the gfx950 path-tracing core behind the C ABI of
include/cpt_amd.h.  This module is plumbing only: every call lands in the HIP library, there is
no CPU fallback, and importing it fails loudly when the library has not been built.

The directory name contains a hyphen, so load it with importlib, e.g.

    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("cuda_path_tracer_amd", ".../cuda-path-tracer_amd/__init__.py")
    cpt = importlib.util.module_from_spec(spec); sys.modules[spec.name] = cpt; spec.loader.exec_module(cpt)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CPT_AMD_LIBRARY") or os.path.join(_HERE, "libcpt_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cpt_amd.h")

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                      "(or __graft_entry__.build()); there is no fallback implementation")

lib = C.CDLL(LIB_PATH)

_f32 = np.ctypeslib.ndpointer(np.float32, flags="C")
_u32 = np.ctypeslib.ndpointer(np.uint32, flags="C")
_vp = C.c_void_p


def _sig(name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


_sig("cpt_last_error", C.c_char_p, [])
_sig("cpt_device_count", C.c_int, [])
_sig("cpt_set_device", C.c_int, [C.c_int])
_sig("cpt_malloc", _vp, [C.c_size_t])
_sig("cpt_free", C.c_int, [_vp])
_sig("cpt_memcpy_h2d", C.c_int, [_vp, _vp, C.c_size_t])
_sig("cpt_memcpy_d2h", C.c_int, [_vp, _vp, C.c_size_t])
_sig("cpt_memset", C.c_int, [_vp, C.c_int, C.c_size_t])
_sig("cpt_synchronize", C.c_int, [])
_sig("cpt_memory_info", C.c_int, [_vp, _vp])
_sig("cpt_build_accel", C.c_int, [_vp, _vp, _f32, C.c_uint32])
_sig("cpt_build_accel_ex", C.c_int, [_vp, _vp, _f32, C.c_uint32, _vp, _vp, _vp])
_sig("cpt_build_accel_noalloc", C.c_int, [_vp, _vp, _f32, C.c_uint32])
_sig("cpt_build_accel_noalloc_max_primitives", C.c_uint32, [])
_sig("cpt_builder_create", _vp, [])
_sig("cpt_builder_destroy", None, [_vp])
_sig("cpt_builder_set_sort", C.c_int, [_vp, C.c_int])
_sig("cpt_builder_set_carrier_limit", C.c_int, [_vp, C.c_uint32])
_sig("cpt_builder_build", C.c_int, [_vp, _vp, _vp, _vp, _f32, C.c_uint32, _vp, _vp])
_sig("cpt_builder_build_records", C.c_int, [_vp, _vp, _vp, _vp, _f32, C.c_uint32, C.c_int, _vp, _vp, _vp])
_sig("cpt_scene_create", _vp, [])
_sig("cpt_scene_destroy", None, [_vp])
_sig("cpt_scene_load_obj", C.c_int, [_vp, C.c_char_p, C.c_int])
_sig("cpt_scene_load_obj_cached", C.c_int, [_vp, C.c_char_p, C.c_int, _vp])
_sig("cpt_scene_load_json", C.c_int, [_vp, C.c_char_p])
_sig("cpt_scene_load_xml", C.c_int, [_vp, C.c_char_p])
_sig("cpt_scene_add_object", C.c_int, [_vp, C.c_char_p, _f32, _f32, _f32, C.c_uint32, _u32, C.c_uint32, _f32, _vp])
_sig("cpt_scene_set_camera", C.c_int, [_vp, _f32, _f32, _f32, C.c_float, C.c_uint32, C.c_uint32])
_sig("cpt_scene_num_objects", C.c_uint32, [_vp])
_sig("cpt_scene_object_info", C.c_int, [_vp, C.c_uint32, C.c_char_p, C.c_size_t, _vp, _vp, _vp, _vp])
_sig("cpt_scene_object_copy", C.c_int, [_vp, C.c_uint32, _vp, _vp, _vp, _vp])
_sig("cpt_scene_object_set_trs", C.c_int, [_vp, C.c_uint32, _vp, _vp, _vp])
_sig("cpt_scene_object_set_material", C.c_int, [_vp, C.c_uint32, _f32])
_sig("cpt_scene_object_transforms", C.c_int, [_vp, C.c_uint32, _vp, _vp])
_sig("cpt_scene_camera", C.c_int, [_vp, _vp, _vp])
_sig("cpt_tracer_create", _vp, [_vp, C.c_uint32, C.c_uint32])
_sig("cpt_tracer_destroy", None, [_vp])
_sig("cpt_tracer_set_max_depth", C.c_int, [_vp, C.c_int])
_sig("cpt_tracer_set_channel", C.c_int, [_vp, C.c_int])
_sig("cpt_tracer_set_integrator", C.c_int, [_vp, C.c_int])
_sig("cpt_tracer_set_traversal", C.c_int, [_vp, C.c_int])
_sig("cpt_tracer_set_node_records", C.c_int, [_vp, C.c_int])
_sig("cpt_tracer_set_wavefront_tail", C.c_int, [_vp, C.c_int])
_sig("cpt_tracer_set_profile_build", C.c_int, [_vp, C.c_int])
_sig("cpt_tracer_build_buffers", C.c_int, [_vp])
_sig("cpt_tracer_update", C.c_int, [_vp])
_sig("cpt_tracer_update_timed", C.c_int, [_vp, C.c_uint32, _vp])
_sig("cpt_tracer_update_many", C.c_int, [_vp, C.c_uint32])
_sig("cpt_tracer_set_samples_per_launch", C.c_int, [_vp, C.c_uint32])
_sig("cpt_tracer_timer_begin", C.c_int, [_vp])
_sig("cpt_tracer_timer_end", C.c_int, [_vp, _vp])
_sig("cpt_tracer_reset_accumulation", C.c_int, [_vp])
_sig("cpt_tracer_synchronize", C.c_int, [_vp])
_sig("cpt_tracer_current_spp", C.c_uint32, [_vp])
_sig("cpt_tracer_num_lights", C.c_uint32, [_vp])
_sig("cpt_tracer_render_samples", C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_int, _vp, C.c_uint32, C.c_uint32])
class RenderParams(C.Structure):
    """cpt_render_params: the block kernel::PathTracer::Render takes (src/kernels/integrator/path.cuh:8-20)."""
    _fields_ = [("scene", _vp), ("output", _vp), ("screen_width", C.c_uint32), ("screen_height", C.c_uint32), ("spp", C.c_uint32),
                ("max_depth", C.c_uint32), ("channel", C.c_int)]


_sig("cpt_path_tracer_render", C.c_int, [C.POINTER(RenderParams)])
_sig("cpt_tracer_film_device_ptr", _vp, [_vp])
_sig("cpt_tracer_set_query_kernel", C.c_int, [_vp, C.c_int])
_sig("cpt_tracer_set_scene_in_lds", C.c_int, [_vp, C.c_int])
_sig("cpt_tracer_film_read", C.c_int, [_vp, _f32])
_sig("cpt_tracer_film_save", C.c_int, [_vp, C.c_char_p])
_sig("cpt_tracer_first_hit", C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp])
_sig("cpt_tracer_occlude", C.c_int, [_vp, _vp, C.c_uint64, _vp])
_sig("cpt_tracer_camera_rays", C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp])
_sig("cpt_tracer_get_accel", C.c_int64, [_vp, C.c_int, _vp, _vp, _vp])
_sig("cpt_tracer_counters", C.c_int, [_vp, _vp])
_sig("cpt_tracer_stack_overflows", C.c_int, [_vp, _vp])
_sig("cpt_tracer_reset_counters", C.c_int, [_vp])
_sig("cpt_tracer_build_stage_ms", C.c_int, [_vp, _vp])
_sig("cpt_tracer_build_extra_ms", C.c_int, [_vp, _vp])
_sig("cpt_shard_samples", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp])
_sig("cpt_comm_unique_id", C.c_int, [_vp])
_sig("cpt_comm_create", _vp, [_vp, C.c_int, C.c_int])
_sig("cpt_comm_destroy", None, [_vp])
_sig("cpt_comm_census", C.c_int, [_vp, _vp, _vp, _vp])
_sig("cpt_tracer_reduce_film", C.c_int, [_vp, _vp, C.c_int, C.c_int])
_sig("cpt_multi_create", _vp, [_vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32])
_sig("cpt_multi_destroy", None, [_vp])
_sig("cpt_multi_set_max_depth", C.c_int, [_vp, C.c_int])
_sig("cpt_multi_set_integrator", C.c_int, [_vp, C.c_int])
_sig("cpt_multi_set_traversal", C.c_int, [_vp, C.c_int])
_sig("cpt_multi_set_node_records", C.c_int, [_vp, C.c_int])
_sig("cpt_multi_set_samples_per_launch", C.c_int, [_vp, C.c_uint32])
_sig("cpt_multi_build_buffers", C.c_int, [_vp])
_sig("cpt_multi_render", C.c_int, [_vp, C.c_uint32, _vp])
_sig("cpt_multi_film_read", C.c_int, [_vp, _f32])
_sig("cpt_multi_film_save", C.c_int, [_vp, C.c_char_p])
_sig("cpt_multi_counters", C.c_int, [_vp, _vp])
_sig("cpt_multi_reset_counters", C.c_int, [_vp])
_sig("cpt_multi_num_devices", C.c_uint32, [_vp])
_sig("cpt_selftest_division", C.c_int, [C.c_uint64, C.c_uint32, _vp])
_sig("cpt_stream_copy_rate", C.c_int, [C.c_uint64, C.c_int, _vp])
_sig("cpt_valu_issue_rate", C.c_int, [C.c_int, C.c_int, _vp, _vp, _vp])
_sig("cpt_write_exr", C.c_int, [C.c_char_p, _f32, C.c_uint32, C.c_uint32])
_sig("cpt_image_load", C.c_int, [C.c_char_p, _vp, _vp, _vp, C.c_size_t])
_sig("cpt_image_save_png", C.c_int, [C.c_char_p, _vp, C.c_uint32, C.c_uint32])
_sig("cpt_scene_object_map", C.c_int, [_vp, C.c_uint32, C.c_int, _vp, _vp, _vp, C.c_size_t])


class CptError(RuntimeError):
    pass


def check(rc):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise CptError(lib.cpt_last_error().decode("utf-8", "replace"))
    return rc


def device_count():
    return lib.cpt_device_count()


def memory_info():
    """(free, total) bytes of the current device."""
    free, total = C.c_size_t(0), C.c_size_t(0)
    check(lib.cpt_memory_info(C.byref(free), C.byref(total)))
    return int(free.value), int(total.value)


def require_device():
    if device_count() < 1:
        raise CptError("no HIP device visible: the gfx950 library cannot run here")


class DeviceArray:
    """A typed device allocation owned through the C ABI's memory helpers."""

    def __init__(self, shape, dtype):
        self.shape = tuple(int(x) for x in shape) if isinstance(shape, (tuple, list)) else (int(shape),)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = lib.cpt_malloc(max(self.nbytes, 1))
        if not self.ptr:
            raise CptError(lib.cpt_last_error().decode())

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        if a.nbytes:
            check(lib.cpt_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes))
        return d

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            check(lib.cpt_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes))
        return out

    def zero(self):
        check(lib.cpt_memset(self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr:
            lib.cpt_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def build_accel(leaf_boxes, merged, want_parents=True):
    """kernel::BuildAccel on host arrays: returns nodes (2n-1, 2) uint32, bboxes (2n-1, 8) float32,
    sorted keys (n,) uint64, parents (2n-1,) uint32 and the five stage times in ms."""
    require_device()
    leaf_boxes = np.ascontiguousarray(leaf_boxes, dtype=np.float32).reshape(-1, 8)
    n = len(leaf_boxes)
    nn = 2 * n - 1
    boxes = np.zeros((nn, 8), dtype=np.float32)
    boxes[n - 1:] = leaf_boxes
    d_boxes = DeviceArray.from_host(boxes)
    d_nodes = DeviceArray((nn, 2), np.uint32)
    d_keys = DeviceArray((n,), np.uint64)
    d_par = DeviceArray((nn,), np.uint32)
    d_par.zero()
    stage = np.zeros(5, dtype=np.float32)
    check(lib.cpt_build_accel_ex(d_nodes.ptr, d_boxes.ptr, np.ascontiguousarray(merged, dtype=np.float32), n, d_keys.ptr,
                                 d_par.ptr if want_parents else None, stage.ctypes.data))
    return d_nodes.to_host(), d_boxes.to_host(), d_keys.to_host(), d_par.to_host(), stage


class Scene:
    """Host scene (Scene + loaders of the reference)."""

    def __init__(self):
        self.h = lib.cpt_scene_create()

    def close(self):
        if self.h:
            lib.cpt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_obj(self, path, bsdf_type=2, cache=False):
        """cache=True reads and keeps <path>.cptcache (a binary copy of the parsed scene); self.from_cache tells
        whether the last load came from it."""
        if cache:
            hit = C.c_int(0)
            check(lib.cpt_scene_load_obj_cached(self.h, os.fsencode(path), bsdf_type, C.byref(hit)))
            self.from_cache = bool(hit.value)
        else:
            check(lib.cpt_scene_load_obj(self.h, os.fsencode(path), bsdf_type))
            self.from_cache = False
        return self

    def load_json(self, path):
        check(lib.cpt_scene_load_json(self.h, os.fsencode(path)))
        return self

    def load_xml(self, path):
        check(lib.cpt_scene_load_xml(self.h, os.fsencode(path)))
        return self

    def add_object(self, name, positions, normals, texcoords, indices, material16):
        positions = np.ascontiguousarray(positions, dtype=np.float32)
        normals = np.ascontiguousarray(normals, dtype=np.float32)
        texcoords = np.ascontiguousarray(texcoords, dtype=np.float32)
        indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        check(lib.cpt_scene_add_object(self.h, name.encode(), positions.reshape(-1), normals.reshape(-1), texcoords.reshape(-1),
                                       len(positions.reshape(-1, 3)), indices, len(indices) // 3,
                                       np.ascontiguousarray(material16, dtype=np.float32), None))
        return self

    def set_camera(self, pos, look_at, up, fov, width, height):
        f = lambda v: np.ascontiguousarray(v, dtype=np.float32)
        check(lib.cpt_scene_set_camera(self.h, f(pos), f(look_at), f(up), float(fov), int(width), int(height)))
        return self

    def num_objects(self):
        return lib.cpt_scene_num_objects(self.h)

    def object(self, i):
        name = C.create_string_buffer(256)
        nv, nt = C.c_uint32(), C.c_uint32()
        mat = np.zeros(16, dtype=np.float32)
        tr = np.zeros(16, dtype=np.float32)
        check(lib.cpt_scene_object_info(self.h, i, name, 256, C.byref(nv), C.byref(nt), mat.ctypes.data, tr.ctypes.data))
        pos = np.zeros((nv.value, 3), dtype=np.float32)
        nrm = np.zeros((nv.value, 3), dtype=np.float32)
        uv = np.zeros((nv.value, 2), dtype=np.float32)
        idx = np.zeros(3 * nt.value, dtype=np.uint32)
        check(lib.cpt_scene_object_copy(self.h, i, pos.ctypes.data, nrm.ctypes.data, uv.ctypes.data, idx.ctypes.data))
        return dict(name=name.value.decode(), positions=pos, normals=nrm, texcoords=uv, indices=idx, material=mat, transform=tr)

    def objects(self):
        return [self.object(i) for i in range(self.num_objects())]

    def object_map(self, index, which):
        """RGBA8 map of the object's material as (h, w) uint32, or None: which = 0 diffuse, 1 specular, 2 emission."""
        w, h = C.c_uint32(0), C.c_uint32(0)
        check(lib.cpt_scene_object_map(self.h, index, which, C.byref(w), C.byref(h), None, 0))
        if w.value == 0:
            return None
        out = np.zeros((h.value, w.value), dtype=np.uint32)
        check(lib.cpt_scene_object_map(self.h, index, which, C.byref(w), C.byref(h), out.ctypes.data, out.size))
        return out

    def set_trs(self, index, translate, rotate_degrees, scale):
        t, r, s = (np.ascontiguousarray(v, dtype=np.float32) for v in (translate, rotate_degrees, scale))
        check(lib.cpt_scene_object_set_trs(self.h, index, t.ctypes.data, r.ctypes.data, s.ctypes.data))
        return self

    def set_material(self, index, material16):
        check(lib.cpt_scene_object_set_material(self.h, index, np.ascontiguousarray(material16, dtype=np.float32)))
        return self

    def object_transforms(self, index):
        m = np.zeros(16, dtype=np.float32)
        inv = np.zeros(16, dtype=np.float32)
        check(lib.cpt_scene_object_transforms(self.h, index, m.ctypes.data, inv.ctypes.data))
        return m, inv

    def camera(self):
        cam = np.zeros(10, dtype=np.float32)
        res = np.zeros(2, dtype=np.uint32)
        check(lib.cpt_scene_camera(self.h, cam.ctypes.data, res.ctypes.data))
        return dict(pos=cam[0:3].copy(), look_at=cam[3:6].copy(), up=cam[6:9].copy(), fov=float(cam[9]), width=int(res[0]),
                    height=int(res[1]))


class Tracer:
    """PathTracer + Film of the reference on one GPU."""

    def __init__(self, scene, width, height, max_depth=-1):
        require_device()
        self.scene = scene
        self.width, self.height = int(width), int(height)
        self.h = check(lib.cpt_tracer_create(scene.h, self.width, self.height))
        check(lib.cpt_tracer_set_max_depth(self.h, max_depth))

    def close(self):
        if self.h:
            lib.cpt_tracer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_max_depth(self, d):
        check(lib.cpt_tracer_set_max_depth(self.h, d))

    def set_integrator(self, mode):
        check(lib.cpt_tracer_set_integrator(self.h, mode))

    def set_traversal(self, mode):
        """0 exact (the reference's order and arithmetic, bit-exact), 1 fast (ordered traversal)."""
        check(lib.cpt_tracer_set_traversal(self.h, mode))

    def set_node_records(self, mode):
        """1: two-level node records are built and walked by the exact traversal (default); 0: binary records only.  Before build_buffers."""
        check(lib.cpt_tracer_set_node_records(self.h, mode))

    def set_query_kernel(self, mode):
        """1: large batches of first-hit queries (exact traversal) run the persistent kernel that refills its lanes; 0 (default): one ray per lane."""
        check(lib.cpt_tracer_set_query_kernel(self.h, mode))

    def set_scene_in_lds(self, on):
        """Whether the small-scene bounce loop reads the records from its LDS copy (default) or from memory; returns whether the copy exists and is used."""
        return check(lib.cpt_tracer_set_scene_in_lds(self.h, 1 if on else 0)) == 1

    def set_wavefront_tail(self, bounces):
        check(lib.cpt_tracer_set_wavefront_tail(self.h, bounces))

    def set_channel(self, c):
        check(lib.cpt_tracer_set_channel(self.h, c))

    def build_buffers(self, profile=False):
        check(lib.cpt_tracer_set_profile_build(self.h, 1 if profile else 0))
        check(lib.cpt_tracer_build_buffers(self.h))
        return self

    def update(self, n=1):
        """n more samples per pixel (cpt_tracer_update_many: same film as n single updates)."""
        if n == 1:
            check(lib.cpt_tracer_update(self.h))
        else:
            check(lib.cpt_tracer_update_many(self.h, n))

    def set_samples_per_launch(self, samples):
        check(lib.cpt_tracer_set_samples_per_launch(self.h, samples))

    def update_timed(self, n):
        """n Update calls; returns the device milliseconds measured on the tracer's stream."""
        ms = C.c_float(0.0)
        check(lib.cpt_tracer_update_timed(self.h, n, C.byref(ms)))
        return ms.value

    def timer_begin(self):
        check(lib.cpt_tracer_timer_begin(self.h))

    def timer_end(self):
        ms = C.c_float(0.0)
        check(lib.cpt_tracer_timer_end(self.h, C.byref(ms)))
        return ms.value

    def reset(self):
        check(lib.cpt_tracer_reset_accumulation(self.h))

    def synchronize(self):
        check(lib.cpt_tracer_synchronize(self.h))

    def render_one(self, d_output, width, height, spp, max_depth=0xFFFFFFFF, channel=0):
        """kernel::PathTracer::Render: one sample (number spp, 1-based) into the device buffer, finished on return."""
        p = RenderParams(self.h, d_output, width, height, spp, max_depth & 0xFFFFFFFF, channel)
        check(lib.cpt_path_tracer_render(C.byref(p)))

    def reduce_film(self, comm, root=0, finish=True):
        """Sum of every rank's film (rendered with render_samples in sum mode) into `root`'s, over RCCL, on the
        tracer's stream; with `finish` the root's film becomes the mean (alpha 1)."""
        check(lib.cpt_tracer_reduce_film(self.h, comm.h, root, 1 if finish else 0))

    def render_samples(self, spp_begin, count, sum_mode=True, d_output=None, width=0, height=0):
        check(lib.cpt_tracer_render_samples(self.h, spp_begin, count, 1 if sum_mode else 0, d_output, width, height))

    def film(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        check(lib.cpt_tracer_film_read(self.h, out.reshape(-1)))
        return out

    def save(self, path):
        check(lib.cpt_tracer_film_save(self.h, os.fsencode(path)))

    def first_hit(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = len(rays)
        d_rays = DeviceArray.from_host(rays)
        d_inst, d_prim = DeviceArray((n,), np.uint32), DeviceArray((n,), np.uint32)
        d_t, d_uv = DeviceArray((n,), np.float32), DeviceArray((n, 2), np.float32)
        check(lib.cpt_tracer_first_hit(self.h, d_rays.ptr, n, d_inst.ptr, d_prim.ptr, d_t.ptr, d_uv.ptr))
        self.synchronize()
        return d_inst.to_host(), d_prim.to_host(), d_t.to_host(), d_uv.to_host()

    def occlude(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = len(rays)
        d_rays = DeviceArray.from_host(rays)
        d_out = DeviceArray((n,), np.uint8)
        check(lib.cpt_tracer_occlude(self.h, d_rays.ptr, n, d_out.ptr))
        self.synchronize()
        return d_out.to_host()

    def camera_rays(self, width, height, spp=1):
        d = DeviceArray((width * height, 8), np.float32)
        check(lib.cpt_tracer_camera_rays(self.h, width, height, spp, d.ptr))
        self.synchronize()
        return d.to_host()

    def accel(self, mesh):
        n = check(lib.cpt_tracer_get_accel(self.h, mesh, None, None, None))
        nodes = np.zeros((2 * n - 1, 2), dtype=np.uint32)
        boxes = np.zeros((2 * n - 1, 8), dtype=np.float32)
        keys = np.zeros(n, dtype=np.uint64)
        check(lib.cpt_tracer_get_accel(self.h, mesh, nodes.ctypes.data, boxes.ctypes.data, keys.ctypes.data))
        return nodes, boxes, keys

    def counters(self):
        out = np.zeros(5, dtype=np.uint64)
        check(lib.cpt_tracer_counters(self.h, out.ctypes.data))
        return dict(rays=int(out[0]), node_visits=int(out[1]), tri_tests=int(out[2]), paths=int(out[3]), shades=int(out[4]))

    def stack_overflows(self):
        """Traversal stack entries dropped since the last counter reset (0 = every traversal was complete)."""
        out = C.c_uint64(0)
        check(lib.cpt_tracer_stack_overflows(self.h, C.byref(out)))
        return int(out.value)

    def reset_counters(self):
        check(lib.cpt_tracer_reset_counters(self.h))

    def build_stage_ms(self):
        out = np.zeros(5, dtype=np.float32)
        check(lib.cpt_tracer_build_stage_ms(self.h, out.ctypes.data))
        extra = np.zeros(2, dtype=np.float32)
        check(lib.cpt_tracer_build_extra_ms(self.h, extra.ctypes.data))
        stages = dict(zip(["morton", "sort", "hierarchy_workgroup", "hierarchy_memory", "reserved"], out.tolist()))
        stages["leaf_boxes"], stages["pack_records"] = float(extra[0]), float(extra[1])
        stages["to_traversable"] = float(out.sum() + extra.sum())
        return stages


def shard_samples(total, rank, world, first=1):
    """(begin, count) of the block of `total` samples that shard `rank` of `world` renders."""
    b, n = C.c_uint32(0), C.c_uint32(0)
    check(lib.cpt_shard_samples(total, rank, world, first, C.byref(b), C.byref(n)))
    return int(b.value), int(n.value)


class Comm:
    """One rank of an RCCL communicator on the current device (one process per GPU)."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(lib.cpt_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, unique_id, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self.h = lib.cpt_comm_create(buf, rank, world)
        if not self.h:
            raise CptError(lib.cpt_last_error().decode("utf-8", "replace"))
        self.rank, self.world = rank, world

    def census(self):
        """(ranks, rank, device) as RCCL reports them for this communicator."""
        n, r, d = C.c_int(0), C.c_int(-1), C.c_int(-1)
        check(lib.cpt_comm_census(self.h, C.byref(n), C.byref(r), C.byref(d)))
        return n.value, r.value, d.value

    def close(self):
        if self.h:
            lib.cpt_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiTracer:
    """One render split over several shards (host thread + stream + replicated build each), summed over RCCL."""

    def __init__(self, scene, width, height, shard_devices, max_depth=-1):
        require_device()
        self.scene = scene
        self.width, self.height = int(width), int(height)
        devs = (C.c_int * len(shard_devices))(*shard_devices)
        self.h = lib.cpt_multi_create(scene.h, self.width, self.height, devs, len(shard_devices))
        if not self.h:
            raise CptError(lib.cpt_last_error().decode("utf-8", "replace"))
        check(lib.cpt_multi_set_max_depth(self.h, max_depth))

    def close(self):
        if self.h:
            lib.cpt_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_integrator(self, mode):
        check(lib.cpt_multi_set_integrator(self.h, mode))

    def set_traversal(self, mode):
        check(lib.cpt_multi_set_traversal(self.h, mode))

    def set_node_records(self, mode):
        check(lib.cpt_multi_set_node_records(self.h, mode))

    def set_samples_per_launch(self, samples):
        check(lib.cpt_multi_set_samples_per_launch(self.h, samples))

    def build_buffers(self):
        check(lib.cpt_multi_build_buffers(self.h))
        return self

    def render(self, total_spp):
        """Samples 1 .. total_spp of every pixel; returns the wall milliseconds of the call (reduce included)."""
        ms = C.c_float(0.0)
        check(lib.cpt_multi_render(self.h, total_spp, C.byref(ms)))
        return ms.value

    def film(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        check(lib.cpt_multi_film_read(self.h, out.reshape(-1)))
        return out

    def save(self, path):
        check(lib.cpt_multi_film_save(self.h, os.fsencode(path)))

    def counters(self):
        out = np.zeros(6, dtype=np.uint64)
        check(lib.cpt_multi_counters(self.h, out.ctypes.data))
        return dict(rays=int(out[0]), node_visits=int(out[1]), tri_tests=int(out[2]), paths=int(out[3]), shades=int(out[4]),
                    stack_overflows=int(out[5]))

    def reset_counters(self):
        check(lib.cpt_multi_reset_counters(self.h))

    def num_devices(self):
        return lib.cpt_multi_num_devices(self.h)


def write_exr(path, rgba):
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    h, w = rgba.shape[0], rgba.shape[1]
    check(lib.cpt_write_exr(os.fsencode(path), rgba.reshape(-1), w, h))


def load_image(path):
    """PNG / PPM / PGM file -> (h, w) uint32 RGBA8 (red in the low byte)."""
    w, h = C.c_uint32(0), C.c_uint32(0)
    check(lib.cpt_image_load(os.fsencode(path), C.byref(w), C.byref(h), None, 0))
    out = np.zeros((h.value, w.value), dtype=np.uint32)
    check(lib.cpt_image_load(os.fsencode(path), C.byref(w), C.byref(h), out.ctypes.data, out.size))
    return out


def save_png(path, rgba):
    rgba = np.ascontiguousarray(rgba, dtype=np.uint32)
    check(lib.cpt_image_save_png(os.fsencode(path), rgba.ctypes.data, rgba.shape[1], rgba.shape[0]))
