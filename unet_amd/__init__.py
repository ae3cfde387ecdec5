"""unet_amd -- MI355X-native CondUNet1D hot path (hand-written HIP kernels behind a C ABI).

Public surface mirrors the reference's ``models.py`` for the path in scope:
``CondUNet1D`` plus its block classes; ``unet_amd.ops`` exposes the kernel-level autograd ops.
"""
from .model import (CondUNet1D, ConvBlock, UpBlock, ModernConvBlock, ModernUpBlock, MultiScaleDilatedConv1d,
                    DilatedConv1d, SqueezeExcitation1D, FiLM, SelfAttention1D, SessionAdaptiveScaling)
from .cond import SpectroTemporalEncoder
from .augment import NoiseAugmentor

__all__ = ["SpectroTemporalEncoder", "NoiseAugmentor", "CondUNet1D", "ConvBlock", "UpBlock", "ModernConvBlock", "ModernUpBlock", "MultiScaleDilatedConv1d",
           "DilatedConv1d", "SqueezeExcitation1D", "FiLM", "SelfAttention1D", "SessionAdaptiveScaling"]
