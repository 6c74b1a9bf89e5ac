"""Convex collision meshes as link hulls (tmx_problem_desc::link_hull).  tesseract loads the convex meshes a URDF names - e.g. the PR2's
trajopt_common/data/pr2/meshes/*/convex/*_convex.stla|stlb|obj - as tesseract::geometry::ConvexMesh; the device path takes their
vertices as a hull in the link frame (contacts by GJK / EPA on the support function, include/tmx_gjk.h: only vertices are needed)."""
import struct

import numpy as np


def load_mesh_vertices(path: str) -> np.ndarray:
    """vertices [n][3] of an ASCII STL (.stl / .stla), binary STL (.stlb) or Wavefront OBJ file, duplicates removed"""
    with open(path, "rb") as f:
        raw = f.read()
    low = path.lower()
    pts = []
    if low.endswith(".obj"):
        for line in raw.decode("latin-1").splitlines():
            w = line.split()
            if len(w) >= 4 and w[0] == "v":
                pts.append([float(w[1]), float(w[2]), float(w[3])])
    elif raw[:5].lower() == b"solid" and b"facet" in raw[:1024]:
        for line in raw.decode("latin-1").splitlines():
            w = line.split()
            if len(w) == 4 and w[0] == "vertex":
                pts.append([float(w[1]), float(w[2]), float(w[3])])
    else:   # binary STL: 80-byte header, uint32 triangle count, 50 bytes per triangle (normal, three vertices, attribute)
        n = struct.unpack_from("<I", raw, 80)[0]
        if 84 + 50 * n > len(raw):
            raise ValueError(f"{path}: not a binary STL")
        for i in range(n):
            v = struct.unpack_from("<12f", raw, 84 + 50 * i)
            pts += [list(v[3:6]), list(v[6:9]), list(v[9:12])]
    if not pts:
        raise ValueError(f"{path}: no vertices found")
    return np.unique(np.round(np.asarray(pts, dtype=np.float64), 9), axis=0)


def convex_hull_vertices(points: np.ndarray, max_vertices: int = 0) -> np.ndarray:
    """the extreme points of a cloud (Qhull); with max_vertices > 0 the hull is thinned to that many vertices by farthest-point
    sampling (an INNER approximation of the hull: report it if you use it - the contact distances grow by at most the spacing)"""
    from scipy.spatial import ConvexHull
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    hv = pts[np.unique(ConvexHull(pts).vertices)]
    if max_vertices and len(hv) > max_vertices:
        keep = [int(np.argmax(np.linalg.norm(hv - hv.mean(axis=0), axis=1)))]
        d = np.linalg.norm(hv - hv[keep[0]], axis=1)
        while len(keep) < max_vertices:
            k = int(np.argmax(d))
            keep.append(k)
            d = np.minimum(d, np.linalg.norm(hv - hv[k], axis=1))
        hv = hv[sorted(keep)]
    return hv


def hull_link(link: int, vertices, radius: float = 0.0, transform=None):
    """a link primitive (link, centre, radius, ("hull", vertices)) for Robot.link_spheres; `transform` (3x4 or 4x4, link_T_mesh)
    places the mesh in the link frame"""
    v = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    if transform is not None:
        T = np.asarray(transform, dtype=np.float64)
        v = v @ T[:3, :3].T + T[:3, 3]
    return (link, (0.0, 0.0, 0.0), float(radius), ("hull", v))
