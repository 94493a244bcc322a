import numpy as np
from scipy.spatial import ConvexHull


def convex_hull(tm):
    from . import Trimesh

    pts = np.asarray(tm.vertices, dtype=np.float64)
    hull = ConvexHull(pts)
    vid = hull.vertices
    remap = {int(v): i for i, v in enumerate(vid)}
    verts = pts[vid]
    centre = verts.mean(0)
    faces = []
    for simplex, eq in zip(hull.simplices, hull.equations):
        f = [remap[int(s)] for s in simplex]
        a, b, c = verts[f[0]], verts[f[1]], verts[f[2]]
        if np.dot(np.cross(b - a, c - a), eq[:3]) < 0:
            f = [f[0], f[2], f[1]]
        faces.append(f)
    return Trimesh(vertices=verts, faces=np.array(faces), process=False)
