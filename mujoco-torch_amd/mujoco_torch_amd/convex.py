"""Convex tables for box and mesh geoms (host side, setup time).

The convex narrow phase (reference ``_src/collision_convex.py``) consumes four per-geom tables
that the reference derives in ``_src/mesh.py:405-447`` (``mesh.get``): hull vertices ``vert[V,3]``,
polygonal faces ``face[F,K]`` (vertex ids, counter-clockwise seen from outside, short faces padded
by repeating their last vertex), unit ``facenormal[F,3]`` and the unique edges ``edge[E,2]``.
Boxes go through the same route as meshes there (corner points -> hull -> merged quads), so they do
here.

Procedure (own implementation on scipy's qhull; the reference uses ``trimesh`` 4.11 on the same
qhull): 3-D hull of the points -> outward-oriented triangles -> union adjacent coplanar triangles
-> each union becomes one polygon ordered by a 2-D hull in the dominant-axis projection -> pad ->
normals from (v1-v0) x (vlast-v0) -> edges shared by >= 2 faces in first-appearance order.

Face ORDER (and the polygon start vertex) follow qhull's output order; the reference's own tests
pin topology only (``test/mesh_test.py:46-67``), so the order is this build's
(SURVEY section 8c "face-order parity unpinned").  ``tests/test_oracle_golden.py`` checks these
tables against the ones the reference's ``mesh.get`` produced when it was run on the stub
``trimesh`` of ``oracle/ref_stubs``.
"""

from __future__ import annotations

from collections import OrderedDict

import numpy as np

from ._enums import GeomType

MAX_FACE_VERTS = 20  # mesh.py:32 _MAX_HULL_FACE_VERTICES
_COPLANAR_COS = 1.0 - 1e-8

_BOX_CORNERS = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float64)


def hull_triangles(points: np.ndarray):
    """3-D hull: (verts[V,3] in input order, tris[T,3] outward-oriented, in qhull simplex order)."""
    from scipy.spatial import ConvexHull

    pts = np.asarray(points, dtype=np.float64)
    hull = ConvexHull(pts)
    vid = np.sort(hull.vertices)
    local = -np.ones(len(pts), dtype=np.int64)
    local[vid] = np.arange(len(vid))
    verts = pts[vid]
    tris = local[hull.simplices]
    for i, eq in enumerate(hull.equations):
        a, b, c = verts[tris[i]]
        if np.dot(np.cross(b - a, c - a), eq[:3]) < 0:
            tris[i] = tris[i][[0, 2, 1]]
    return verts, tris


def _tri_normals(verts, tris):
    a = verts[tris]
    n = np.cross(a[:, 1] - a[:, 0], a[:, 2] - a[:, 0])
    return n / np.linalg.norm(n, axis=1, keepdims=True)


def coplanar_groups(verts, tris):
    """Unions of >= 2 edge-adjacent coplanar triangles, ordered by their smallest triangle id."""
    normals = _tri_normals(verts, tris)
    owner = {}
    for ti, t in enumerate(tris):
        for k in range(3):
            e = (int(min(t[k], t[(k + 1) % 3])), int(max(t[k], t[(k + 1) % 3])))
            owner.setdefault(e, []).append(ti)
    root = list(range(len(tris)))

    def find(x):
        while root[x] != x:
            root[x] = root[root[x]]
            x = root[x]
        return x

    for ts in owner.values():
        if len(ts) == 2 and float(np.dot(normals[ts[0]], normals[ts[1]])) > _COPLANAR_COS:
            a, b = find(ts[0]), find(ts[1])
            if a != b:
                root[max(a, b)] = min(a, b)
    groups = OrderedDict()
    for ti in range(len(tris)):
        groups.setdefault(find(ti), []).append(ti)
    merged = [(np.array(g), normals[g[0]]) for _, g in sorted(groups.items()) if len(g) > 1]
    return merged


def _polygon(verts, ids, normal):
    """Orders the vertex ids of one planar facet counter-clockwise about ``normal``."""
    from scipy.spatial import ConvexHull

    drop = int(np.argmax(np.abs(normal)))
    keep = [a for a in range(3) if a != drop]
    ring = ConvexHull(verts[ids][:, keep]).vertices  # counter-clockwise in the (keep[0], keep[1]) plane
    flip = (normal[drop] > 0) != (drop != 1)  # (x, z) is a left-handed pair seen from +y
    return ids[ring[::-1] if flip else ring]


def polygon_faces(verts, tris):
    """Triangles that belong to no coplanar union first, then one polygon per union; padded."""
    merged = coplanar_groups(verts, tris)
    used = set(int(t) for g, _ in merged for t in g)
    polys = []
    for g, normal in merged:
        ids = np.unique(tris[g])
        poly = _polygon(verts, ids, normal)
        if len(poly) > MAX_FACE_VERTS:
            poly = poly[:: len(poly) // MAX_FACE_VERTS + 1]
        polys.append(poly)
    loose = [tris[t] for t in range(len(tris)) if t not in used]
    width = max([len(p) for p in polys] + ([3] if loose or not polys else []))
    pad = lambda f: np.concatenate([f, np.full(width - len(f), f[-1], dtype=f.dtype)])
    return np.array([pad(np.asarray(f, dtype=np.int64)) for f in loose + polys], dtype=np.int64)


def face_normals(verts, face):
    fv = verts[face]
    n = np.cross(fv[:, 1] - fv[:, 0], fv[:, -1] - fv[:, 0])
    return n / np.linalg.norm(n, axis=1).reshape(-1, 1)


def unique_edges(face):
    """Vertex-id pairs shared by at least two faces, in first-appearance order (mesh.py:57-90)."""
    count = OrderedDict()
    for f in face:
        for k in range(len(f)):
            a, b = int(f[k]), int(f[k - 1])
            if a != b:
                e = (min(a, b), max(a, b))
                count[e] = count.get(e, 0) + 1
    return np.array([e for e, c in count.items() if c >= 2], dtype=np.int64).reshape(-1, 2)


def tables_from_points(points) -> dict:
    verts, tris = hull_triangles(points)
    face = polygon_faces(verts, tris)
    return dict(vert=verts, face=face, facenormal=face_normals(verts, face), edge=unique_edges(face))


def geom_convex_tables(m) -> list:
    """One table dict (or None) per geom: boxes from their scaled corners, meshes from mesh_vert."""
    out, cache = [], {}
    for g in range(int(m.ngeom)):
        t = int(m.geom_type[g])
        if t == GeomType.BOX:
            pts = _BOX_CORNERS * np.asarray(m.geom_size[g], dtype=np.float64).reshape(1, 3)
        elif t == GeomType.MESH and int(m.geom_dataid[g]) >= 0:
            i = int(m.geom_dataid[g])
            a = int(m.mesh_vertadr[i])
            pts = np.asarray(m.mesh_vert[a : a + int(m.mesh_vertnum[i])], dtype=np.float64)
        else:
            out.append(None)
            continue
        key = pts.tobytes()
        if key not in cache:
            cache[key] = tables_from_points(pts)
        out.append(cache[key])
    return out


def shape_key(tables, g):
    """Grouping key of a geom in the candidate table (collision_driver.py:149-157)."""
    t = tables[g]
    if t is None:
        return ((-1,), (-1,), (-1,))
    return (tuple(t["face"].shape), tuple(t["vert"].shape), tuple(t["edge"].shape))
