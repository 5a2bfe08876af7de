"""Drop-in for the reference's ``models/models.py`` (LeakyReLU(0.2) + dropout variant)."""
from ._core import (SINTRO, BuildingBlock, ResNetDecoder, ResNetEncoder, ResNetVAE,  # noqa: F401
                    SoftIntroVAE, UpsampleBuildingkBlock, VAEResNetEncoder)

__all__ = ["BuildingBlock", "UpsampleBuildingkBlock", "ResNetEncoder", "ResNetDecoder",
           "VAEResNetEncoder", "ResNetVAE", "SoftIntroVAE"]


# ---- 160x192x160 volumes -------------------------------------------------------------------------
# The reference's scripts build 80x96x80 models only (z-1200main.py:158, aug-z-1200main.py); the
# full-resolution benchmark configuration keeps the same 10x12x10 = 1200 latent by putting one more
# stride-2 stage of 64 channels in front of the z=1200 stack (and its mirror at the end of the decoder).
BLOCKS_Z1200 = [[64, 1, 2], [128, 1, 2], [256, 2, 2]]                 # z-1200main.py:158, 80x96x80 input
BLOCKS_Z1200_160 = [[64, 1, 2], [64, 1, 2], [128, 1, 2], [256, 2, 2]]   # 160x192x160 input


def soft_intro_vae_z1200(full_resolution: bool = False) -> "SoftIntroVAE":
    """``SoftIntroVAE(64, ...)`` with a 1200-dimensional conv latent: the reference's z=1200 model for
    80x96x80 volumes, or (``full_resolution``) its four-stage extension for 160x192x160 volumes."""
    return SoftIntroVAE(64, [list(b) for b in (BLOCKS_Z1200_160 if full_resolution else BLOCKS_Z1200)])
