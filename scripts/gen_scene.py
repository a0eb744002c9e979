"""Writes a synthetic scene in the reference's file formats: <out>.obj + <out>.mtl + <out>.json.
usage: python scripts/gen_scene.py <heightfield|soup> <ntris> <out-prefix> [width height]
The mesh carries v/vt/vn and `f a/a/a` faces (the reference needs normals and uvs in the file), one
diffuse material, plus a small emissive quad (no light, no render: src/pathtracer/pathtracer.cpp:27-29).
Numbers are written with nine significant digits, which round-trips every float32, so the file holds the generator's
triangles exactly; the text is formatted in pieces by up to 16 worker processes (a 10M-triangle scene is about 1.1 GB)."""
import io, json, os, sys
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
import numpy as np
from scenes import synthetic_scene

ROWS_PER_PIECE = 250_000


def _format(piece):
    rows, fmt = piece
    buf = io.StringIO()
    np.savetxt(buf, rows, fmt=fmt)
    return buf.getvalue()


def write_rows(f, pool, rows, fmt):
    pieces = [(rows[i:i + ROWS_PER_PIECE], fmt) for i in range(0, len(rows), ROWS_PER_PIECE)]
    for text in (pool.imap(_format, pieces) if pool else map(_format, pieces)):
        f.write(text)


def main():
    kind, ntris, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
    H = int(sys.argv[5]) if len(sys.argv) > 5 else 1080
    objs, cam = synthetic_scene(kind, ntris)
    name = os.path.basename(out)
    with open(out + ".mtl", "w") as f:
        f.write("newmtl ground\nKd 0.7 0.7 0.7\nKs 0 0 0\nNs 1\n\nnewmtl lamp\nKd 0 0 0\nKs 0 0 0\nKe 17 12 4\n")
    pool = Pool(min(16, os.cpu_count() or 1)) if ntris > 2 * ROWS_PER_PIECE else None
    with open(out + ".obj", "w") as f:
        f.write(f"mtllib {name}.mtl\n")
        base = 1
        for o, mtl in zip(objs, ("ground", "lamp")):
            f.write(f"o {o['name']}\nusemtl {mtl}\n")
            write_rows(f, pool, o["positions"], "v %.9g %.9g %.9g")
            write_rows(f, pool, o["texcoords"], "vt %.9g %.9g")
            write_rows(f, pool, o["normals"], "vn %.9g %.9g %.9g")
            idx = o["indices"].reshape(-1, 3).astype(np.int64) + base
            write_rows(f, pool, np.repeat(idx, 3, axis=1), "f %d/%d/%d %d/%d/%d %d/%d/%d")
            base += len(o["positions"])
    if pool:
        pool.close()
    json.dump({"camera": {"pos": [0.5, 0.3, 0.1], "look_at": [0.5, 0.0, 0.45], "up": [0, 1, 0], "fov": 40.0, "resolution": [W, H]}},
              open(out + ".json", "w"))
    print("wrote", out + ".obj", os.path.getsize(out + ".obj") >> 20, "MiB")


if __name__ == "__main__":
    main()
