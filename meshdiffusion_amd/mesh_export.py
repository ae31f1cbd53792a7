"""`.npy` sampled grids -> `.obj` meshes, and fitted DMTet dicts -> training grids: the host-side data formats on
either side of the hot path, without nvdiffrast / pytorch3d / pymeshlab.

* `samples_to_obj`: what the tail of the reference's `nvdiffrec/eval.py:385-447` does with the sampler's output
  (`main_diffusion.py --mode=uncond_gen` writes `{eval_dir}/{i}.npy`): gather the tet-grid vertices out of the cubic
  grid, sign / clip, marching tetrahedra (the HIP kernel, 32 meshes per launch), write `{idx:06d}.obj`.  The
  reference's render-a-preview and pymeshlab clean-up steps are rendering / post-processing and out of scope.
* `save_obj`: the plain "v x y z" / "f i j k" (1-based) subset of the format `pytorch3d.io.save_obj` writes
  (`eval.py:436-440`).
* `tet_to_grid` / `dicts_to_grids`: `data/tets_to_3dgrid.py:7-49`, the `dmt_dict_{id}.pt` -> `grid_{id}.pt` step that
  produces the training set read by `lib/dataset/shapenet_dmtet_dataset.py`.

    python -m meshdiffusion_amd.mesh_export --sample_path out/0.npy --tet_path 64_tets_cropped.npz --out meshes/
"""
import argparse
import os

import numpy as np
import torch

from .dmtet import GridMesher, tet_vertices_to_grid_index


def save_obj(path, verts, faces, decimal_places=None, normals=None):
    """verts float [V,3], faces int [F,3] (0-based) -> Wavefront OBJ.  normals (optional, float [V,3], one per vertex,
    e.g. dmtet.auto_normals): written as `vn` lines and referenced as `f i//i`."""
    v = torch.as_tensor(verts).detach().cpu().numpy().astype(np.float64)
    f = torch.as_tensor(faces).detach().cpu().numpy().astype(np.int64) + 1
    fmt = "%f" if decimal_places is None else f"%.{int(decimal_places)}f"
    with open(path, "w") as fh:
        for row in v:
            fh.write("v " + " ".join(fmt % c for c in row) + "\n")
        if normals is not None:
            for row in torch.as_tensor(normals).detach().cpu().numpy().astype(np.float64):
                fh.write("vn " + " ".join(fmt % c for c in row) + "\n")
            for row in f:
                fh.write("f %d//%d %d//%d %d//%d\n" % (row[0], row[0], row[1], row[1], row[2], row[2]))
        else:
            for row in f:
                fh.write("f %d %d %d\n" % (row[0], row[1], row[2]))


def load_obj(path):
    """Inverse of save_obj (tests / round trips): returns (verts float32 [V,3], faces int64 [F,3] 0-based)."""
    vs, fs = [], []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                vs.append([float(c) for c in t[1:4]])
            elif t[0] == "vn":
                continue
            elif t[0] == "f":
                fs.append([int(c.split("/")[0]) - 1 for c in t[1:4]])
    return np.asarray(vs, np.float32).reshape(-1, 3), np.asarray(fs, np.int64).reshape(-1, 3)


def samples_to_obj(samples, tet_vertices, tet_indices, out_dir, resolution=None, batch=32, device="cuda", start_index=0,
                   with_normals=False):
    """samples: array [M,4,R,R,R] (or a path to the sampler's .npy).  Writes {out_dir}/{i:06d}.obj, returns their paths.
    with_normals: also write the smooth vertex normals the reference computes right after extraction
    (eval.py:422 `mesh.auto_normals`; its own OBJ export, eval.py:436-440, drops them)."""
    if isinstance(samples, (str, os.PathLike)):
        samples = np.load(samples)
    samples = np.asarray(samples)
    R = int(resolution or samples.shape[-1])
    mesher = GridMesher(tet_vertices, tet_indices, R, device=device)
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for lo in range(0, samples.shape[0], batch):
        meshes = mesher(torch.from_numpy(samples[lo:lo + batch]))
        for k, (verts, faces, _face_tet) in enumerate(meshes):
            path = os.path.join(out_dir, "{:06d}.obj".format(start_index + lo + k))
            nrm = None
            if with_normals and verts.shape[0] > 0:
                from .dmtet import auto_normals
                nrm = auto_normals(verts, faces)[0]
            save_obj(path, verts, faces, normals=nrm)
            paths.append(path)
    return paths


def tet_to_grid(grid_index, sdf, deform, resolution):
    """data/tets_to_3dgrid.py:7-15: scatter per-tet-vertex SDF [N] and deformation [N,3] into a [4,R,R,R] grid."""
    idx = torch.as_tensor(grid_index).long()
    grid = torch.zeros(4, resolution, resolution, resolution)
    grid[0, idx[:, 0], idx[:, 1], idx[:, 2]] = torch.as_tensor(sdf, dtype=torch.float32).reshape(-1)
    grid[1:, idx[:, 0], idx[:, 1], idx[:, 2]] = torch.as_tensor(deform, dtype=torch.float32).transpose(0, 1)
    return grid


def dicts_to_grids(tet_vertices, src_dir, dst_dir, resolution, indices):
    """data/tets_to_3dgrid.py:17-49: `{src_dir}/dmt_dict_{i:05d}.pt` ({'sdf', 'deform'}) -> `{dst_dir}/grid_{i:05d}.pt`."""
    idx = tet_vertices_to_grid_index(tet_vertices)
    os.makedirs(dst_dir, exist_ok=True)
    written = []
    for i in indices:
        src = os.path.join(src_dir, "dmt_dict_{:05d}.pt".format(i))
        if not os.path.exists(src):
            continue
        d = torch.load(src, map_location="cpu", weights_only=False)
        dst = os.path.join(dst_dir, "grid_{:05d}.pt".format(i))
        torch.save(tet_to_grid(idx, d["sdf"], d["deform"], resolution), dst)
        written.append(dst)
    return written


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--sample_path", required=True, help=".npy written by --mode=uncond_gen / cond_gen")
    ap.add_argument("--tet_path", required=True, help="<R>_tets_cropped.npz (vertices, indices)")
    ap.add_argument("--out", required=True, help="directory for the .obj files")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--normals", action="store_true", help="write smooth vertex normals (vn) next to the positions")
    a = ap.parse_args(argv)
    tet = np.load(a.tet_path)
    paths = samples_to_obj(a.sample_path, tet["vertices"], tet["indices"], a.out, batch=a.batch, with_normals=a.normals)
    print(f"wrote {len(paths)} meshes to {a.out}")


if __name__ == "__main__":
    main()
