"""Drop-in for utils/stylegan2/op/__init__.py: the same three names, backed by the prebuilt
gfx950 library instead of JIT-compiled CUDA."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d
