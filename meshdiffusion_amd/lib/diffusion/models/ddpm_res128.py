"""DDPM 3-D U-Net for 128^3 x 4 DMTet grids on MI355X -- drop-in for the reference's
lib/diffusion/models/ddpm_res128.py (`DDPMRes128`, __init__ :43-135, forward :137-215).

Differences from res64 (SURVEY.md 8a "res128 deltas"): the stem, `mask_layer`, `pos_layer` and the head
are 5x5x5 / pad 2 convolutions (:90-92, :132), there is no `coords` input (`use_coords = False`, :77; the
`pos_layer` parameters still exist in the checkpoint), and level 0 always has 2 residual blocks (:98, :118).
The reference's config names the model 'ddpm_res128_v2' (configs/res128.py:40), which it never
registers; both names resolve here.
"""
from . import utils
from .ddpm_res64 import DDPMUNet3D


@utils.register_model(name="ddpm_res128")
class DDPMRes128(DDPMUNet3D):
    KSIZE, USE_COORDS, LEVEL0_BLOCKS = 5, False, 2


utils._MODELS.setdefault("ddpm_res128_v2", DDPMRes128)
