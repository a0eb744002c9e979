"""Seeded generators of the synthetic scenes the measurements and the parity tests use (numpy only): a random triangle
soup, a tessellated heightfield, a small emissive quad, and the leaf boxes of a triangle list as
src/scene/mesh.cpp:88-98 fills them.  Every rank of a multi-GPU run builds the same scene from the same seed."""
import numpy as np


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


# The camera the measurements use on the synthetic scenes: low over the terrain (inside the cube of the soup), so that
# nearly every ray of a 16:9 film hits.  Overrides pos / look_at / fov of the camera synthetic_scene returns.
BENCH_CAMERA = dict(pos=np.array([0.5, 0.3, 0.1], dtype=np.float32), look_at=np.array([0.5, 0.0, 0.45], dtype=np.float32), fov=40.0)


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
