"""TransHE -- ViT-tiny over the N_c canonical-body tokens, HIP-backed.

Host-side mirror of /root/reference/lib/networks/vision_transformer.py:257-407
(`Attention`, `Block`, `VisionTransformer`, `vit_tiny`): same module tree and
parameter names (``blocks.{i}.norm1/attn.qkv/attn.proj/norm2/mlp.fc1/mlp.fc2``,
``norm``, ``cls_token``, ``mask_token``, ``PE._freqs/_phases``) so reference
checkpoints load with ``strict=True``; same call signature
``forward(x[V,N,C], PE[V,N,3], mask=None) -> [V,N,C]``.

The arithmetic runs in transhuman_amd/csrc/k_vit.hip through the C-ABI
(``th_vit_forward``).  There is no torch fallback: without the HIP library the
call raises.
"""
import torch
from torch import nn

from .encoder import _PEBuffers


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class VisionTransformer(nn.Module):
    def __init__(self, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4.0):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.depth = depth
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))     # unused with mask=None
        self.mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        assert embed_dim % 6 == 0
        self.PE = _PEBuffers(embed_dim // 6, include_input=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        nn.init.trunc_normal_(self.mask_token, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.constant_(m.bias, 0)
        self._pe_cache = None

    def get_PE(self, PE, device=None, out_shape=None):
        """Sin-cos table of the normalised canonical centres.

        32 octaves reach pi*2^31: a handful of ulps in the argument flips the
        value, so the table is evaluated with torch's own addcmul+sin exactly
        as the reference does (vision_transformer.py:131-132) -- on the host,
        once, since it depends only on constants (hard part 1).
        """
        key = (PE.data_ptr(), tuple(PE.shape), str(PE.device), PE._version)
        if self._pe_cache is not None and self._pe_cache[0] == key:
            return self._pe_cache[1]
        V, N, _ = PE.shape
        x = PE.detach().to("cpu", torch.float32).flatten(0, 1)
        tab = self.PE.cpu()(x).view(V, N, -1).to(PE.device)
        self.PE.to(PE.device)
        self._pe_cache = (key, tab)
        return tab

    def forward(self, x, PE, mask=None, graph=False):
        from .. import hip
        if mask is not None and bool(mask.sum() != 0):
            x = x.clone()
            x[mask] = self.mask_token.to(x.dtype)
        pe = self.get_PE(PE)
        return hip.vit_forward(self, x, pe, graph=graph)


def vit_tiny(depth, **kw):
    return VisionTransformer(embed_dim=192, depth=depth, num_heads=3, mlp_ratio=4, **kw)
