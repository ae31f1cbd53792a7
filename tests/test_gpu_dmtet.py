"""Marching tetrahedra: HIP kernel vs the reference's golden hashes (bit-exact indices AND verts)
and vs the numpy oracle, on the shipped 64-grid (BASELINE config #5 sizes: 32 meshes per call)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLD

pytestmark = pytest.mark.gpu


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def tet():
    t = np.load(os.path.join(GOLD, "64_tets_cropped.npz"))
    return t["vertices"], t["indices"]


def test_dmtet_bit_exact_vs_reference_golden(hip_lib, tet):
    from meshdiffusion_amd.dmtet import DMTet
    from oracle.gen_golden import dmtet_cases
    verts, idx = tet
    gold = np.load(os.path.join(GOLD, "dmtet.npz"))
    pos, cases = dmtet_cases(verts)
    dm = DMTet()
    tets_t = torch.as_tensor(idx, dtype=torch.long).cuda()
    for name, sdf in cases.items():
        v, f, uvs, uv_idx, ftet, vvi = dm(pos.cuda(), sdf.cuda(), tets_t)
        assert f.dtype == torch.int64
        assert (v.shape[0], f.shape[0]) == tuple(gold[f"{name}_counts"]), name
        assert _sha(f.cpu().numpy()) == str(gold[f"{name}_faces_sha"]), f"{name}: faces differ"
        assert _sha(v.cpu().numpy()) == str(gold[f"{name}_verts_sha"]), f"{name}: verts differ"
        assert _sha(uv_idx.cpu().numpy()) == str(gold[f"{name}_uv_idx_sha"]), name
        assert _sha(vvi.cpu().numpy()) == str(gold[f"{name}_vvi_sha"]), name
        assert uvs.dtype == torch.float32 and tuple(uvs.shape) == tuple(gold[f"{name}_uvs_shape"]), name
        assert _sha(uvs.cpu().numpy()) == str(gold[f"{name}_uvs_sha"]), f"{name}: uvs differ"
        assert ftet.dtype == torch.int64 and _sha(ftet.cpu().numpy()) == str(gold[f"{name}_ftet_sha"]), f"{name}: face_to_valid_tet differs"
        # smooth vertex normals (mesh.py:200-229 `auto_normals`): face normals equal to the reference up to fma contraction (1 ulp), vertex normals up
        # to the summation order of the splat (float atomics here, sequential scatter_add in the reference on the CPU)
        from meshdiffusion_amd.dmtet import auto_normals
        from oracle import dmtet_oracle
        vn, fnrm = auto_normals(v, f)
        fh = gold[f"{name}_fnrm_head"]
        assert np.abs(fnrm.cpu().numpy()[:256] - fh).max() <= 1e-6 * np.abs(fh).max(), f"{name}: face normals differ"
        assert np.abs(fnrm.double().sum(0).cpu().numpy() - gold[f"{name}_fnrm_sum"]).max() < 1e-6 * max(1.0, np.abs(gold[f"{name}_fnrm_sum"]).max())
        ok = dmtet_oracle.well_conditioned_normals(v.cpu().numpy(), f.cpu().numpy())     # all but a few vertices of the noisy cases
        vnp = vn.cpu().numpy()
        assert ok.mean() > 0.85 and np.abs(vnp[:256] - gold[f"{name}_vnrm_head"])[ok[:256]].max() < 1e-4
        assert np.abs((vnp.astype(np.float64) * ok[:, None]).sum(0) - gold[f"{name}_vnrm_sum"]).max() < 1e-2
        vn_or, _ = dmtet_oracle.auto_normals(v.cpu().numpy(), f.cpu().numpy())
        assert np.abs(vnp - vn_or)[ok].max() < 1e-4 and np.abs(np.linalg.norm(vnp, axis=1) - 1).max() < 1e-5


def test_dmtet_batch32_vs_oracle(hip_lib, tet):
    """32 meshes in one launch from synthetic sampled grids (sign SDF + clipped deformation)."""
    from meshdiffusion_amd.dmtet import GridMesher
    from oracle import dmtet_oracle
    verts, idx = tet
    g = torch.Generator().manual_seed(11)
    M, R = 32, 64
    ax = torch.linspace(-1, 1, R)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    grids = torch.empty(M, 4, R, R, R)
    for m in range(M):
        c = torch.rand(3, generator=g) * 0.4 - 0.2
        rad = 0.3 + 0.3 * float(torch.rand(1, generator=g))
        grids[m, 0] = rad - ((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2).sqrt() + 0.05 * torch.sin(9 * X + m)
        grids[m, 1:] = torch.randn(3, R, R, R, generator=g) * 0.7
    mesher = GridMesher(verts, idx, R)
    meshes = mesher(grids.cuda())
    assert len(meshes) == M
    for m in (0, 7, 31):
        pos, sdf = dmtet_oracle.grid_to_tet_inputs(grids[m].numpy(), verts, mesh_scale=2.1, deform_scale=2.0, R=R)
        vo, fo, fto = dmtet_oracle.marching_tets(pos, sdf, idx)
        v, f, ft = meshes[m]
        assert np.array_equal(f.cpu().numpy(), fo) and np.array_equal(ft.cpu().numpy(), fto)
        assert np.abs(v.cpu().numpy() - vo).max() <= 1e-6


def test_npy_samples_to_obj_files(hip_lib, tmp_path):
    """`.npy` -> `.obj` (eval.py:385-447 without renderer / pymeshlab): files parse back to exactly the meshes the
    batched HIP marching tets returns, which test_marching_tets_* pin bit-exactly to the reference."""
    import os
    from conftest import GOLD
    from meshdiffusion_amd import mesh_export
    from meshdiffusion_amd.dmtet import GridMesher
    tet = np.load(os.path.join(GOLD, "64_tets_cropped.npz"))
    g = torch.Generator().manual_seed(2)
    ax = torch.linspace(-1, 1, 64)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    grids = torch.empty(3, 4, 64, 64, 64)
    for m in range(3):
        grids[m, 0] = 0.35 + 0.05 * m - (X ** 2 + Y ** 2 + Z ** 2).sqrt()
        grids[m, 1:] = torch.randn(3, 64, 64, 64, generator=g) * 0.5
    npy = tmp_path / "0.npy"
    np.save(npy, grids.numpy())
    paths = mesh_export.samples_to_obj(str(npy), tet["vertices"], tet["indices"], str(tmp_path / "meshes"), batch=2)
    assert [os.path.basename(p) for p in paths] == ["000000.obj", "000001.obj", "000002.obj"]
    meshes = GridMesher(tet["vertices"], tet["indices"], 64)(grids)
    for p, (v, f, _) in zip(paths, meshes):
        v2, f2 = mesh_export.load_obj(p)
        assert np.array_equal(f2, f.cpu().numpy()) and f2.shape[0] > 1000
        assert np.abs(v2 - v.cpu().numpy()).max() < 1e-5          # %f keeps 6 decimals
    mesh_export.main(["--sample_path", str(npy), "--tet_path", os.path.join(GOLD, "64_tets_cropped.npz"), "--out", str(tmp_path / "cli")])
    assert len(os.listdir(tmp_path / "cli")) == 3


def test_dmtet_degenerate_sdfs_vs_oracle(hip_lib, tet):
    """Edge cases of the extraction: no crossing at all (all-positive / all-negative SDF: the reference returns empty
    verts / faces / uv_idx / face_to_valid_tet / valid_vert_idx and the full uv table) and a single inside vertex."""
    from meshdiffusion_amd.dmtet import DMTet, auto_normals
    from oracle import dmtet_oracle
    from oracle.gen_golden import dmtet_cases
    verts, idx = tet
    pos, _ = dmtet_cases(verts)
    tets_t = torch.as_tensor(idx, dtype=torch.long).cuda()
    n = len(verts)
    dm = DMTet()
    for name, sdf in (("all_pos", torch.ones(n)), ("all_neg", -torch.ones(n)),
                      ("one_vertex", torch.where(torch.arange(n) == 12345, torch.ones(()), -torch.ones(())))):
        v, f, uvs, uv_idx, ftet, vvi = dm(pos.cuda(), sdf.cuda(), tets_t)
        vo, fo, fto = dmtet_oracle.marching_tets(pos.numpy(), sdf.numpy(), idx)
        assert tuple(v.shape) == vo.shape and tuple(f.shape) == fo.shape and f.dtype == torch.int64, name
        assert np.array_equal(f.cpu().numpy(), fo) and np.array_equal(v.cpu().numpy(), vo) and np.array_equal(ftet.cpu().numpy(), fto), name
        assert tuple(uv_idx.shape) == (fo.shape[0], 3) and uvs.shape[1] == 2 and uvs.shape[0] > 0, name
        if name == "one_vertex":
            assert (v.shape[0], f.shape[0], vvi.shape[0]) == (14, 24, 15)        # what the unmodified reference returns
            vn, _ = auto_normals(v, f)
            assert bool(torch.isfinite(vn).all())
        else:
            assert v.shape[0] == 0 and f.shape[0] == 0 and vvi.numel() == 0 and ftet.numel() == 0
