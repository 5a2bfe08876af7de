"""Drop-in for the reference's ``models/models.py`` (LeakyReLU(0.2) + dropout variant)."""
from ._core import (SINTRO, BuildingBlock, ResNetDecoder, ResNetEncoder, ResNetVAE,  # noqa: F401
                    SoftIntroVAE, UpsampleBuildingkBlock, VAEResNetEncoder)

__all__ = ["BuildingBlock", "UpsampleBuildingkBlock", "ResNetEncoder", "ResNetDecoder",
           "VAEResNetEncoder", "ResNetVAE", "SoftIntroVAE"]
