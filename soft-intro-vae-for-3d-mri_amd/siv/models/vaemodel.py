"""Drop-in for the reference's ``models/vaemodel.py``: the same stacks with nn.ReLU and no
dropout (models/vaemodel.py:8-134, 202-230)."""
from . import _core
from ._core import PLAIN


class BuildingBlock(_core.BuildingBlock):
    def __init__(self, in_ch, out_ch, stride, bias=False):
        super().__init__(in_ch, out_ch, stride, bias, PLAIN)


class UpsampleBuildingkBlock(_core.UpsampleBuildingkBlock):
    def __init__(self, in_ch, out_ch, stride, bias=False):
        super().__init__(in_ch, out_ch, stride, bias, PLAIN)


class ResNetEncoder(_core.ResNetEncoder):
    def __init__(self, in_ch, block_setting):
        super().__init__(in_ch, block_setting, PLAIN)


class VAEResNetEncoder(_core.VAEResNetEncoder):
    def __init__(self, in_ch, block_setting):
        super().__init__(in_ch, block_setting, PLAIN)


ResNetDecoder = _core.ResNetDecoder


class ResNetVAE(_core.ResNetVAE):
    variant = PLAIN
