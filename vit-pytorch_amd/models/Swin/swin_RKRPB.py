"""Swin Transformer with RKR factors shared across tasks and specialised by piggyback masks on the vitk HIP kernels (MI355X / gfx950).

Drop-in for ``/root/reference/models/Swin/swin_RKRPB.py``, which is ``swin_RKR.py`` with ``ElementWiseRG`` / ``ElementWiseSFG`` in place of
``RG_FC`` / ``SFG_FC`` (``Mlp`` :14-45, ``WindowAttention`` :80-170); every other class is the same and so
it is here: the classes of ``swin_RKR.py`` with the layer types swapped.  The window attention, the
shift folded into its addressing, patch merging and stochastic depth are shared; the per-task weight
operand of each GEMM comes from the layer's ``effective(task)`` as in ``models/ViT/modeling_RKRPB.py``.
"""
from . import swin_RKR as rkr
from .swin_RKR import (DropPath, PatchEmbed, PatchMerging, to_2tuple, trunc_normal_, window_partition,  # noqa: F401
                       window_reverse)
from models.ViT.layers_PB import ElementWiseRG, ElementWiseSFG  # noqa: F401


class _Layers:
    @staticmethod
    def fc(config, h_in, h_out):
        return ElementWiseRG(h_in, h_out, config)

    @staticmethod
    def sfg(config, c_out):
        return ElementWiseSFG(c_out, config) if config.SFG else None


class Mlp(rkr.Mlp):
    layers = _Layers


class WindowAttention(rkr.WindowAttention):
    layers = _Layers


class SwinTransformerBlock(rkr.SwinTransformerBlock):
    attention_cls, mlp_cls = WindowAttention, Mlp


class BasicLayer(rkr.BasicLayer):
    block_cls = SwinTransformerBlock


class SwinTransformer(rkr.SwinTransformer):
    layer_cls = BasicLayer
