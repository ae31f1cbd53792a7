"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's marching tetrahedra
(nvdiffrec/lib/geometry/dmtet.py:105-163 `DMTet.__call__`, LUTs :34-54, sort_edges :60-68)
and of the grid->tet-vertex gather of nvdiffrec/eval.py:412-419.

Follows the reference's run-time algorithm literally (sort + unique over the edges of the valid
tets) so that it is an independent check of the product's static-edge-table formulation.
Pinned against the imported reference by oracle/gen_golden.py (fixtures in tests/golden/).
"""
import numpy as np

TRIANGLE_TABLE = np.array([
    [-1, -1, -1, -1, -1, -1], [1, 0, 2, -1, -1, -1], [4, 0, 3, -1, -1, -1], [1, 4, 2, 1, 3, 4],
    [3, 1, 5, -1, -1, -1], [2, 3, 0, 2, 5, 3], [1, 4, 0, 1, 5, 4], [4, 2, 5, -1, -1, -1],
    [4, 5, 2, -1, -1, -1], [4, 1, 0, 4, 5, 1], [3, 2, 0, 3, 5, 2], [1, 3, 5, -1, -1, -1],
    [4, 1, 2, 4, 3, 1], [3, 0, 4, -1, -1, -1], [2, 0, 1, -1, -1, -1], [-1, -1, -1, -1, -1, -1]],
    dtype=np.int64)
NUM_TRIANGLES = np.array([0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0], dtype=np.int64)
BASE_TET_EDGES = np.array([0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3], dtype=np.int64)


def marching_tets(pos, sdf, tets):
    """pos [N,3] f32, sdf [N] f32, tets [T,4] int -> verts [V,3] f32, faces [F,3] i64, face_tet [F]."""
    pos = np.asarray(pos, dtype=np.float32)
    sdf = np.asarray(sdf, dtype=np.float32)
    tets = np.asarray(tets, dtype=np.int64)
    occ = sdf > 0
    occ4 = occ[tets.reshape(-1)].reshape(-1, 4)
    occ_sum = occ4.sum(-1)
    valid = (occ_sum > 0) & (occ_sum < 4)
    # dmtet.py:114-116: edges of valid tets, each sorted (min,max), unique rows (lexicographic)
    edges = tets[valid][:, BASE_TET_EDGES].reshape(-1, 2)
    edges = np.stack([edges.min(1), edges.max(1)], -1)
    uniq, inv = np.unique(edges, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    # :118-122: crossing edges get consecutive vertex ids in the sorted order
    cross = occ[uniq.reshape(-1)].reshape(-1, 2).sum(-1) == 1
    mapping = -np.ones(uniq.shape[0], dtype=np.int64)
    mapping[cross] = np.arange(cross.sum(), dtype=np.int64)
    idx_map = mapping[inv].reshape(-1, 6)
    iv = uniq[cross]
    # :124-132: linear interpolation; sdf pair (s0, -s1), weights flip(pair)/sum(pair)
    p = pos[iv.reshape(-1)].reshape(-1, 2, 3)
    s = sdf[iv.reshape(-1)].reshape(-1, 2, 1).copy()
    s[:, 1] *= np.float32(-1)
    den = s.sum(1, keepdims=True)
    w = s[:, ::-1] / den
    verts = (p * w).astype(np.float32)
    verts = verts[:, 0] + verts[:, 1]
    # :136-145: faces, 1-triangle tets first then 2-triangle tets
    tetindex = (occ4[valid] * (2 ** np.arange(4))[None]).sum(-1)
    ntri = NUM_TRIANGLES[tetindex]
    f1 = np.take_along_axis(idx_map[ntri == 1], TRIANGLE_TABLE[tetindex[ntri == 1]][:, :3], axis=1).reshape(-1, 3)
    f2 = np.take_along_axis(idx_map[ntri == 2], TRIANGLE_TABLE[tetindex[ntri == 2]][:, :6], axis=1).reshape(-1, 3)
    faces = np.concatenate([f1, f2], 0)
    gidx = np.arange(tets.shape[0], dtype=np.int64)[valid]
    face_tet = np.concatenate([gidx[ntri == 1], np.repeat(gidx[ntri == 2], 2)], 0)
    return verts, faces, face_tet


def grid_to_tet_inputs(grid, tet_verts, mesh_scale=2.1, deform_scale=2.0, R=64):
    """eval.py:412-419 + dmtet.py:293-304: cubic grid [4,R,R,R] -> (deformed positions, sign sdf)."""
    tet_verts = np.asarray(tet_verts, dtype=np.float32)
    uniq = np.unique(tet_verts)
    dx = uniq[1] - uniq[0]
    idx = np.round((tet_verts - tet_verts.min()) / dx).astype(np.int64)
    g = np.asarray(grid, dtype=np.float32)
    sdf = np.sign(g[0][idx[:, 0], idx[:, 1], idx[:, 2]]).astype(np.float32)
    deform = np.clip(g[1:][:, idx[:, 0], idx[:, 1], idx[:, 2]].T, -1.0, 1.0).astype(np.float32)
    pos = tet_verts * np.float32(mesh_scale) + np.float32(2.0 / (2 * R)) * deform * np.float32(deform_scale)
    return pos.astype(np.float32), sdf


def auto_normals(verts, faces):
    """nvdiffrec/lib/render/mesh.py:200-229 (`auto_normals`) with util.dot / util.safe_normalize (util.py:20-35):
    returns (v_nrm [V,3] float32, f_nrm [F,3] float32).  Sums in face order like torch.scatter_add_ on the CPU."""
    v = np.asarray(verts, np.float32)
    f = np.asarray(faces, np.int64)
    v0, v1, v2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    f_nrm = np.cross(v1 - v0, v2 - v0).astype(np.float32)
    v_nrm = np.zeros_like(v)
    for k in range(3):
        np.add.at(v_nrm, f[:, k], f_nrm)
    d = np.sum(v_nrm * v_nrm, -1, keepdims=True)
    v_nrm = np.where(d > 1e-20, v_nrm, np.array([0.0, 0.0, 1.0], np.float32))
    d = np.sum(v_nrm * v_nrm, -1, keepdims=True)
    return (v_nrm / np.sqrt(np.maximum(d, 1e-20))).astype(np.float32), f_nrm


def well_conditioned_normals(verts, faces, rel=1e-3):
    """Mask [V] of the vertices whose normal is numerically meaningful: the splatted face normals do not cancel (length of
    their sum > rel * the sum of their lengths).  At the others (a handful on noisy SDFs) the direction is decided by the
    last bit of the summation order, so implementations with a different order (float atomics, fma) legitimately differ."""
    v = np.asarray(verts, np.float32)
    f = np.asarray(faces, np.int64)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]).astype(np.float64)
    s, a = np.zeros((v.shape[0], 3)), np.zeros(v.shape[0])
    for k in range(3):
        np.add.at(s, f[:, k], fn)
        np.add.at(a, f[:, k], np.linalg.norm(fn, axis=1))
    return np.linalg.norm(s, axis=1) > rel * np.maximum(a, 1e-30)
