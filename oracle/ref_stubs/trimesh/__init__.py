"""Container-only stand-in for the slice of ``trimesh`` the reference's mesh.py touches.

TEST INFRASTRUCTURE (oracle/gen_golden.py only).  ``Trimesh(vertices, faces)`` with ``facets`` /
``facets_normal`` (groups of >= 2 adjacent coplanar triangles) and ``convex.convex_hull`` on top of
scipy's qhull.  Face ORDER is this stub's own; the reference's tests pin topology only
(test/mesh_test.py:46-67), so tie-breaks that depend on it are unpinned.
"""
import numpy as np

from . import convex  # noqa: F401


class Trimesh:
    def __init__(self, vertices=None, faces=None, process=True):
        v = np.asarray(vertices, dtype=np.float64)
        f = np.asarray(faces, dtype=np.int64)
        if process and len(v):
            # merge duplicate vertices (trimesh does this by default)
            uv, inv = np.unique(np.round(v, 10), axis=0, return_inverse=True)
            first = np.zeros(len(uv), dtype=np.int64)
            for i in range(len(v) - 1, -1, -1):
                first[inv[i]] = i
            order = np.argsort(first)
            rank = np.empty_like(order)
            rank[order] = np.arange(len(order))
            v = v[first[order]]
            f = rank[inv][f]
        self.vertices = v
        self.faces = f
        self._facets = None

    @property
    def face_normals(self):
        a = self.vertices[self.faces]
        n = np.cross(a[:, 1] - a[:, 0], a[:, 2] - a[:, 0])
        return n / np.linalg.norm(n, axis=1, keepdims=True)

    def _compute_facets(self):
        nf = len(self.faces)
        normals = self.face_normals
        edge_faces = {}
        for fi, f in enumerate(self.faces):
            for k in range(3):
                e = tuple(sorted((int(f[k]), int(f[(k + 1) % 3]))))
                edge_faces.setdefault(e, []).append(fi)
        parent = list(range(nf))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x

        for fs in edge_faces.values():
            if len(fs) == 2 and np.dot(normals[fs[0]], normals[fs[1]]) > 1 - 1e-8:
                a, b = find(fs[0]), find(fs[1])
                if a != b:
                    parent[max(a, b)] = min(a, b)
        groups = {}
        for fi in range(nf):
            groups.setdefault(find(fi), []).append(fi)
        facets = [np.array(g) for _, g in sorted(groups.items()) if len(g) > 1]
        self._facets = facets
        self._facets_normal = np.array([normals[g[0]] for g in facets]) if facets else np.zeros((0, 3))

    @property
    def facets(self):
        if self._facets is None:
            self._compute_facets()
        return self._facets

    @property
    def facets_normal(self):
        if self._facets is None:
            self._compute_facets()
        return self._facets_normal
