"""Static mesh topology tables (host side, numpy, computed once per template).

Everything here is integer index bookkeeping that the reference gets from PyTorch3D `Meshes`
(edges_packed / faces_packed_to_edges_packed), `SubdivideMeshes` (optimize_sequence.py:67-89,
utils/visualize.py:50-54) and `_C.mesh_normal_consistency_find_verts` (optimize_sequence.py:537)
and *recomputes every iteration*; here it is built once and uploaded to HBM as int32 tables.
"""
import numpy as np


def unique_edges(faces, V):
    """Edges in PyTorch3D `edges_packed` order + per-face edge ids.

    Edge list = unique (min,max) pairs sorted by hash V*min+max; face_to_edge[:, k] is the edge
    OPPOSITE corner k (columns e12, e20, e01).  
    """
    faces = np.asarray(faces, np.int64)
    v0, v1, v2 = faces[:, 0], faces[:, 1], faces[:, 2]
    e = np.concatenate([np.stack([v1, v2], 1), np.stack([v2, v0], 1), np.stack([v0, v1], 1)], 0)
    lo, hi = e.min(1), e.max(1)
    h = lo * V + hi
    u, inv = np.unique(h, return_inverse=True)
    edges = np.stack([u // V, u % V], 1)
    face_to_edge = inv.reshape(3, len(faces)).T
    return edges, face_to_edge


def subdivide_topology(faces0, V0):
    """SubdivideMeshes face table: new vertex V0+rank(edge); faces [f0;f1;f2;f3]."""
    edges0, f2e = unique_edges(faces0, V0)
    faces0 = np.asarray(faces0, np.int64)
    fe = f2e + V0
    f0 = np.stack([faces0[:, 0], fe[:, 2], fe[:, 1]], 1)
    f1 = np.stack([faces0[:, 1], fe[:, 0], fe[:, 2]], 1)
    f2 = np.stack([faces0[:, 2], fe[:, 1], fe[:, 0]], 1)
    f3 = fe
    return edges0, np.concatenate([f0, f1, f2, f3], 0)


def csr_from_pairs(rows, cols, n_rows):
    """CSR (offsets, values) with rows ascending and, inside a row, original order kept."""
    rows = np.asarray(rows, np.int64)
    order = np.argsort(rows, kind="stable")
    counts = np.bincount(rows, minlength=n_rows)
    off = np.zeros(n_rows + 1, np.int64)
    np.cumsum(counts, out=off[1:])
    return off.astype(np.int32), np.asarray(cols)[order].astype(np.int32)


def normal_consistency_pairs(faces, V):
    """(P,4) int table [v0, v1, a, b]: for each pair of faces sharing edge (v0,v1), the two opposite
    vertices.  Same enumeration as PyTorch3D mesh_normal_consistency: all pairs among the faces
    incident on an edge (exactly one pair for a manifold interior edge, none for boundary edges)."""
    edges, f2e = unique_edges(faces, V)
    faces = np.asarray(faces, np.int64)
    inc = [[] for _ in range(len(edges))]
    for f in range(len(faces)):
        for k in range(3):
            inc[f2e[f, k]].append(faces[f, k])      # corner k is opposite edge f2e[f,k]
    out = []
    for e, opp in enumerate(inc):
        for i in range(len(opp)):
            for j in range(i + 1, len(opp)):
                out.append((edges[e, 0], edges[e, 1], opp[i], opp[j]))
    return np.asarray(out, np.int32).reshape(-1, 4)
