"""Training-grid dataset -- host-side mirror of the reference's lib/dataset/shapenet_dmtet_dataset.py:9-56.

Each item is one `[4, r, r, r]` fp32 cubic grid (channel 0 = SDF, 1..3 = deformation) stored as `.pt`
(`data/tets_to_3dgrid.py:49`) or `.npy`, listed by a JSON file of paths; an optional JSON list of integer ids
(parsed from `..._<id>.<ext>`) filters it.  Behaviour kept as the reference has it, quirks included:

* `normalize_sdf` replaces `datum[:, :1]` -- i.e. the FIRST DEPTH SLAB of all four channels, not channel 0 --
  by its sign with 0 -> +1 (reference :39-42; the stored SDFs of the published pipeline are already +-1);
* augmentation adds one uniform(-0.5, 0.5)*0.01 offset per deformation channel to the non-empty cells, scaled
  by `resolution / r` when the stored grid is smaller than the model grid, then multiplies by the grid mask
  (cropped to r^3), and finally zero-pads up to the model resolution at the high end of every axis.

RNG: the augmentation uses the process-global torch CPU generator like the reference (`torch.rand(3)`).
"""
import json

import numpy as np
import torch
from torch.utils.data import Dataset


class ShapeNetDMTetDataset(Dataset):
    def __init__(self, root, grid_mask, deform_scale=1.0, aug=False, filter_meta_path=None, normalize_sdf=True,
                 extension="pt"):
        super().__init__()
        with open(root, "r") as f:
            self.fpath_list = json.load(f)
        if extension not in ("pt", "npy"):
            raise AssertionError(f"extension must be 'pt' or 'npy', got {extension!r}")
        self.deform_scale, self.normalize_sdf, self.aug, self.extension = deform_scale, normalize_sdf, aug, extension
        self.grid_mask = grid_mask.detach().cpu()
        self.resolution = self.grid_mask.size(-1)
        if filter_meta_path is not None:
            with open(filter_meta_path, "r") as f:
                keep = set(json.load(f))
            cut = len(extension) + 1
            ids = [int(p.rstrip().split("_")[-1][:-cut]) for p in self.fpath_list]
            self.fpath_list = [p for p, i in zip(self.fpath_list, ids) if i in keep]

    def __len__(self):
        return len(self.fpath_list)

    def _read(self, path):
        if self.extension == "pt":
            return torch.load(path, map_location="cpu", weights_only=False)
        return torch.from_numpy(np.load(path)).clone()

    def __getitem__(self, idx):
        with torch.no_grad():
            datum = self._read(self.fpath_list[idx])
            r, R = datum.size(-1), self.resolution
            if self.normalize_sdf:
                sign = torch.sign(datum[:, :1])
                sign[sign == 0] = 1.0
                datum[:, :1] = sign
            if self.aug:
                live = datum[1:].abs().sum(dim=0, keepdim=True) != 0
                datum[1:] = datum[1:] + (torch.rand(3)[:, None, None, None] - 0.5) * 0.01 * live / (r / R)
                gm = self.grid_mask[0]
                datum = datum * (gm[:, :r, :r, :r] if r < R else gm)
            if r < R:
                d = R - r
                datum = torch.nn.functional.pad(datum, (0, d, 0, d, 0, d, 0, 0))
        return datum
