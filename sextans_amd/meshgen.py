"""Matrices whose GRAPH has locality but whose NUMBERING does not follow a grid: the inputs that defeat a stride-based row
clustering and need the graph clustering of csrc/graph_cluster.hip.  Test / measurement infrastructure (numpy + scipy on the
host), not part of the reference and not on any product path.

  permute_symmetric      P A P^T of a CSR matrix (rows and columns renumbered together, columns ascending per row)
  node_permutation       a seeded random renumbering of the nodes of a multi-dof mesh matrix (dof rows of a node stay together)
  rcm_node_permutation   reverse Cuthill-McKee of the node graph (scipy), the classic bandwidth-reducing order of FEM packages
  jittered_mesh3d        an unstructured "Delaunay-like" mesh: jittered points, edges between points closer than a radius
                         (variable degree, no stencil), numbered along a sweep direction, randomly, or by RCM
"""
import numpy as np


def _csr(M, K, rp, ci, v):
    import scipy.sparse as sp
    return sp.csr_matrix((np.asarray(v, np.float32), np.asarray(ci, np.int32), np.asarray(rp, np.int32)), shape=(M, K))


def permute_symmetric(rp, ci, v, M, new_of_old):
    """Row / column i of the input becomes row / column new_of_old[i]; returns (rp, ci, v) with ascending columns."""
    A = _csr(M, M, rp, ci, v)
    new_of_old = np.asarray(new_of_old, np.int64)
    old_of_new = np.empty(M, np.int64)
    old_of_new[new_of_old] = np.arange(M)
    B = A[old_of_new]                       # rows in the new order
    B = B.tocsr()
    B.indices = new_of_old[B.indices].astype(np.int32)   # columns relabelled
    B.has_sorted_indices = False
    B.sort_indices()
    return B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data.astype(np.float32)


def expand_dof(node_new_of_old, dof):
    n = np.asarray(node_new_of_old, np.int64)
    return (n[:, None] * dof + np.arange(dof, dtype=np.int64)[None, :]).reshape(-1)


def node_permutation(nnodes, dof, seed):
    """new_of_old for the ROWS of a matrix with `dof` consecutive rows per node: nodes shuffled, a node's rows stay together."""
    return expand_dof(np.random.RandomState(seed).permutation(nnodes), dof)


def node_graph(rp, ci, M, dof):
    """Pattern of the node graph (scipy CSR, M / dof nodes) of a matrix with dof consecutive rows per node."""
    import scipy.sparse as sp
    nn = M // dof
    rows = np.repeat(np.arange(M, dtype=np.int64), np.diff(rp)) // dof
    cols = np.asarray(ci, np.int64) // dof
    G = sp.csr_matrix((np.ones(len(cols), np.int32), (rows, cols)), shape=(nn, nn))
    G.sum_duplicates()
    return G


def rcm_node_permutation(rp, ci, M, dof):
    """new_of_old for the rows: reverse Cuthill-McKee of the node graph."""
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    G = node_graph(rp, ci, M, dof)
    order = reverse_cuthill_mckee(G, symmetric_mode=True)        # order[i] = old node at new position i
    new_of_old = np.empty(len(order), np.int64)
    new_of_old[order] = np.arange(len(order))
    return expand_dof(new_of_old, dof)


def jittered_mesh3d(nx, ny, nz, seed, numbering="sweep", radius=1.55, dof=1):
    """Points p(i,j,k) = (i,j,k) + U(-0.45, 0.45)^3; an edge joins two points of neighbouring cells closer than `radius`
    (plus the diagonal): ~14 neighbours per point on average, between 6 and 24 -- no two rows share a stencil.  Values U(-1, 1).
    numbering: "sweep" = by ascending x coordinate (what an advancing-front mesher writes: locality in one direction only),
    "random", "rcm", or "grid" (the cell order).  dof > 1: dense dof x dof blocks.  Returns (rp, ci, v, M)."""
    import scipy.sparse as sp
    rs = np.random.RandomState(seed)
    n = nx * ny * nz
    idx = np.arange(n).reshape(nz, ny, nx)
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    P = np.stack([x, y, z], -1).astype(np.float64) + rs.uniform(-0.45, 0.45, (nz, ny, nx, 3))
    rows, cols = [np.arange(n)], [np.arange(n)]
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if (dz, dy, dx) <= (0, 0, 0):
                    continue                                    # each unordered pair once
                zs = slice(max(0, -dz), nz - max(0, dz)); zt = slice(max(0, dz), nz - max(0, -dz))
                ys = slice(max(0, -dy), ny - max(0, dy)); yt = slice(max(0, dy), ny - max(0, -dy))
                xs = slice(max(0, -dx), nx - max(0, dx)); xt = slice(max(0, dx), nx - max(0, -dx))
                d = np.linalg.norm(P[zs, ys, xs] - P[zt, yt, xt], axis=-1)
                m = d < radius
                a, b = idx[zs, ys, xs][m], idx[zt, yt, xt][m]
                rows += [a, b]; cols += [b, a]
    r = np.concatenate(rows); c = np.concatenate(cols)
    G = sp.csr_matrix((np.ones(len(r), np.float32), (r, c)), shape=(n, n))
    if numbering == "sweep":
        order = np.argsort(P.reshape(-1, 3)[:, 0], kind="stable")
    elif numbering == "random":
        order = rs.permutation(n)
    elif numbering == "rcm":
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        order = reverse_cuthill_mckee(G, symmetric_mode=True)
    elif numbering == "grid":
        order = np.arange(n)
    else:
        raise ValueError(numbering)
    new_of_old = np.empty(n, np.int64)
    new_of_old[order] = np.arange(n)
    G = G[order]
    G.indices = new_of_old[G.indices].astype(np.int32)
    G.has_sorted_indices = False
    G.sort_indices()
    if dof > 1:
        G = sp.kron(G, np.ones((dof, dof), np.float32), format="csr")
        G.sort_indices()
    M = n * dof
    v = rs.uniform(-1, 1, G.nnz).astype(np.float32)
    return G.indptr.astype(np.int32), G.indices.astype(np.int32), v, M
