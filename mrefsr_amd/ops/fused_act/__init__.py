from .fused_act import FusedLeakyReLU, fused_leaky_relu

__all__ = ['FusedLeakyReLU', 'fused_leaky_relu']
