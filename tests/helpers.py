"""Shared test plumbing: everything oracle/bindings.py offers (the checker and its scene-file reader), the seeded scene
generators of the package (cuda-path-tracer_amd/scenes.py) and a few scene variants only the tests need.  Nothing here is
imported by the product path or by bench.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
from bindings import *  # noqa: E402,F401,F403
from bindings import GOLDEN, SCENES, ORACLE_DIR  # noqa: E402,F401
from scenes import *  # noqa: E402,F401,F403


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


