"""transhuman_amd -- MI355X-native (gfx950) implementation of TransHuman's
volumetric rendering hot path behind the reference's lib/networks API.

    from transhuman_amd.networks.make_network import make_network
    from transhuman_amd.networks.renderer.make_renderer import make_renderer

The arithmetic lives in transhuman_amd/libtranshuman_hip.so (hand-written HIP,
C ABI in include/transhuman_hip.h); build it with `python -m transhuman_amd.build`.
"""
__version__ = "0.1.0"
