"""Marching tetrahedra on MI355X -- host side of md_marching_tets.

Drop-in for the reference's nvdiffrec/lib/geometry/dmtet.py `DMTet.__call__` :105-163
(same signature and 6-tuple result) plus the grid -> tet-vertex gather of
nvdiffrec/eval.py:389-419 / `get_deformed` dmtet.py:293-304 and the grid-mask construction of
data/get_tet_mask.py:9-37.

Static preprocessing (once per tet grid, torch ops): the lexicographically sorted unique edge list
of ALL tets and the [T,6] tet->edge-id table.  Per call: four launches (chunked over edges / tets) for M meshes.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .hip_ops import _ptr, _stream

BASE_TET_EDGES = (0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3)


class TetTables:
    """Static tables of one tet grid, resident on the device."""

    def __init__(self, tets, device):
        tets = torch.as_tensor(tets).to(device=device, dtype=torch.int64)
        be = torch.tensor(BASE_TET_EDGES, dtype=torch.int64, device=device)
        e = tets[:, be].reshape(-1, 2)
        e = torch.stack([e.min(dim=1).values, e.max(dim=1).values], dim=-1)
        uniq, inv = torch.unique(e, dim=0, return_inverse=True)  # rows sorted lexicographically
        self.n_tets, self.n_edges = tets.shape[0], uniq.shape[0]
        self.tets = tets.to(torch.int32).contiguous()
        self.edges = uniq.to(torch.int32).contiguous()
        self.tet_edges = inv.reshape(-1, 6).to(torch.int32).contiguous()
        self.tets64 = tets


class MeshCounts:
    """Per-mesh (vertices, faces, 1-triangle tets, 2-triangle tets) of one md_marching_tets call: int32 [M, 4], copied to a pinned
    host buffer ASYNCHRONOUSLY on the launch stream; the first host read waits for that copy's event.  Behaves like the numpy
    array (indexing, np.asarray, .sum through it)."""

    def __init__(self, dev_counts):
        self.device_counts = dev_counts                       # consumers on the GPU can read the sizes without any host sync
        self._host = torch.empty(dev_counts.shape, dtype=torch.int32, pin_memory=True)
        self._host.copy_(dev_counts, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()
        self._np = None

    def numpy(self):
        if self._np is None:
            self._event.synchronize()
            self._np = self._host.numpy()
        return self._np

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, k):
        return self.numpy()[k]

    def __len__(self):
        return self._host.shape[0]

    @property
    def shape(self):
        return tuple(self._host.shape)


class MeshBatch:
    """The M meshes of one md_marching_tets call as a lazy sequence of (verts [V,3], faces [F,3] int64, face_tet [F]) views into
    the call's output buffers (allocated at their static bounds V <= E, F <= 2T): the trimming -- which needs the counts on the
    host -- happens at the first access, not inside the call, so a caller that queues more GPU work does not stall on it."""

    def __init__(self, verts, faces, face_tet, counts):
        self.verts, self.faces, self.face_tet, self.counts = verts, faces, face_tet, counts

    def __len__(self):
        return self.verts.shape[0]

    def __getitem__(self, m):
        if isinstance(m, slice):
            return [self[i] for i in range(*m.indices(len(self)))]
        if m < 0:
            m += len(self)
        if not 0 <= m < len(self):
            raise IndexError(m)
        c = self.counts.numpy()
        nv, nf = int(c[m, 0]), int(c[m, 1])
        return self.verts[m, :nv], self.faces[m, :nf], self.face_tet[m, :nf]

    def __iter__(self):
        return (self[m] for m in range(len(self)))


_MT_WORKSPACE = {}


def _mt_workspace(nbytes, device):
    """The call's scratch (edge -> vertex ids, chunk offsets) is dead when its last kernel ends: one buffer per (device, stream),
    grown to the largest request -- launches on a stream are ordered, so consecutive calls can share it."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _MT_WORKSPACE.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _MT_WORKSPACE[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return buf


def release_workspace():
    """Drop the per-stream scratch buffers of md_marching_tets (the next call allocates again)."""
    _MT_WORKSPACE.clear()


def marching_tets_batch(pos, sdf, tables):
    """pos [M,N,3] f32, sdf [M,N] f32 on the GPU -> (MeshBatch, MeshCounts): meshes[m] = (verts [V,3], faces [F,3] int64,
    face_tet [F]), counts[m] = (V, F, 1-triangle tets, 2-triangle tets).  No host synchronisation inside the call: the counts
    travel to a pinned buffer behind the kernels and are awaited at the first read."""
    lib = _lib.load()
    if not pos.is_cuda:
        raise _lib.MeshDiffusionHipError("marching tets runs on the GPU only (no CPU fallback)")
    pos = pos.to(torch.float32).contiguous()
    sdf = sdf.to(torch.float32).contiguous()
    M, N = sdf.shape
    E, T = tables.n_edges, tables.n_tets
    dev = pos.device
    verts = torch.empty((M, E, 3), dtype=torch.float32, device=dev)
    faces = torch.empty((M, 2 * T, 4), dtype=torch.int64, device=dev)     # [..., :3] faces, then M * 2T face -> tet ids behind them
    face_tet = faces.view(-1)[M * 2 * T * 3:].view(M, 2 * T)
    faces = faces.view(-1)[:M * 2 * T * 3].view(M, 2 * T, 3)
    counts = torch.empty((M, 4), dtype=torch.int32, device=dev)
    ws_bytes = lib.md_marching_tets_workspace_bytes(M, E, T)
    ws = _mt_workspace(ws_bytes, dev)
    _lib.check(lib.md_marching_tets(_ptr(pos), _ptr(sdf), _ptr(tables.tets), _ptr(tables.edges),
                                    _ptr(tables.tet_edges), M, N, E, T, _ptr(verts), _ptr(faces),
                                    _ptr(face_tet), _ptr(counts), _ptr(ws), ws_bytes, _stream()),
               "md_marching_tets")
    cnt = MeshCounts(counts)
    return MeshBatch(verts, faces, face_tet, cnt), cnt


def _map_uv(face_tet, n1, num_tets, device):
    """Per-tet texture atlas (dmtet.py:70-99): face_gidx = 2*tet (+1 for a tet's second triangle)."""
    max_idx = num_tets * 2
    Nn = int(np.ceil(np.sqrt((max_idx + 1) // 2)))
    lin = torch.linspace(0, 1 - (1 / Nn), Nn, dtype=torch.float32, device=device)
    tex_y, tex_x = torch.meshgrid(lin, lin, indexing="ij")
    pad = 0.9 / Nn
    uvs = torch.stack([tex_x, tex_y, tex_x + pad, tex_y, tex_x + pad, tex_y + pad, tex_x, tex_y + pad],
                      dim=-1).view(-1, 2)
    F = face_tet.shape[0]
    tri_idx = torch.zeros(F, dtype=torch.int64, device=device)
    if F > n1:
        tri_idx[n1:] = torch.arange(F - n1, device=device) % 2
    tet_idx = face_tet  # (face_gidx // 2); _idx(t, N) = (t // N) * N + t % N = t
    uv_idx = torch.stack((tet_idx * 4, tet_idx * 4 + tri_idx + 1, tet_idx * 4 + tri_idx + 2), dim=-1).view(-1, 3)
    return uvs, uv_idx


class DMTet:
    """`DMTet()(pos_nx3, sdf_n, tet_fx4) -> (verts, faces, uvs, uv_idx, face_to_valid_tet, valid_vert_idx)`."""

    def __init__(self):
        self._tables = {}

    def tables_for(self, tet_fx4):
        key = (tet_fx4.data_ptr(), tuple(tet_fx4.shape), str(tet_fx4.device), tet_fx4._version)
        if key not in self._tables:
            self._tables = {key: TetTables(tet_fx4, tet_fx4.device)}
        return self._tables[key]

    def __call__(self, pos_nx3, sdf_n, tet_fx4):
        with torch.no_grad():
            tb = self.tables_for(tet_fx4)
            meshes, cnt = marching_tets_batch(pos_nx3[None], sdf_n[None], tb)
            verts, faces, face_tet = meshes[0]
            uvs, uv_idx = _map_uv(face_tet, int(cnt[0, 2]), tb.n_tets, verts.device)
            tets_used = torch.unique(face_tet)
            valid_vert_idx = tb.tets64[tets_used].long().unique()
            return verts, faces, uvs, uv_idx, face_tet.long(), valid_vert_idx


# ---- cubic grid <-> tet grid ---------------------------------------------------------------------
def tet_vertices_to_grid_index(vertices):
    """eval.py:389-398 / evaler.py:187-195: tet-vertex positions -> integer grid coordinates."""
    v = torch.as_tensor(vertices)
    uniq = v[:].unique()
    dx = uniq[1] - uniq[0]
    return torch.round((v - v.min()) / dx).long()


def grid_mask_from_tets(vertices, R):
    """data/get_tet_mask.py:9-37: 1 where a tet-grid vertex lives, else 0.  float32 [R,R,R]."""
    idx = tet_vertices_to_grid_index(vertices)
    m = torch.zeros(R, R, R)
    m[idx[:, 0], idx[:, 1], idx[:, 2]] = 1.0
    return m


class GridMesher:
    """`.npy` grids [M,4,R,R,R] -> meshes, the eval.py:400-431 path without the renderer."""

    def __init__(self, tet_vertices, tet_indices, R, mesh_scale=2.1, deform_scale=2.0, device="cuda"):
        self.R, self.deform_scale = R, deform_scale
        dev = torch.device(device)
        v = torch.as_tensor(tet_vertices, dtype=torch.float32)
        self.idx = tet_vertices_to_grid_index(v).to(dev)
        self.verts = (v * mesh_scale).to(dev)            # dmtet.py:219
        self.tables = TetTables(torch.as_tensor(tet_indices), dev)

    def inputs(self, grids):
        g = torch.as_tensor(grids).to(self.verts.device, torch.float32)
        i0, i1, i2 = self.idx[:, 0], self.idx[:, 1], self.idx[:, 2]
        sdf = torch.sign(g[:, 0][:, i0, i1, i2])                              # [M,N]
        deform = g[:, 1:][:, :, i0, i1, i2].transpose(1, 2).clip(-1.0, 1.0)     # [M,N,3]
        pos = self.verts[None] + 2 / (self.R * 2) * deform * self.deform_scale  # dmtet.py:303
        return pos.contiguous(), sdf.contiguous()

    def __call__(self, grids):
        pos, sdf = self.inputs(grids)
        meshes, _ = marching_tets_batch(pos, sdf, self.tables)
        return meshes


_ = C


def auto_normals(verts, faces):
    """Smooth vertex normals of a mesh (nvdiffrec/lib/render/mesh.py:200-229, the call at nvdiffrec/eval.py:422 on the
    marching-tets output): returns (v_nrm float32 [V,3], f_nrm float32 [F,3] unnormalised) -- md_vertex_normals."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    if not verts.is_cuda:
        raise _lib.MeshDiffusionHipError("auto_normals runs on the GPU only")
    v = verts.to(torch.float32).contiguous()
    f = faces.to(torch.int64).contiguous()
    v_nrm = torch.empty_like(v)
    f_nrm = torch.empty((f.shape[0], 3), dtype=torch.float32, device=v.device)
    _lib.check(lib.md_vertex_normals(C.c_void_p(v.data_ptr()), C.c_void_p(f.data_ptr()), v.shape[0], f.shape[0],
                                     C.c_void_p(v_nrm.data_ptr()), C.c_void_p(f_nrm.data_ptr()),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "md_vertex_normals")
    return v_nrm, f_nrm
