"""NeRF sin-cos embedders -- /root/reference/lib/networks/embedder.py:4-55.

Only the view embedder (view_res = 4 -> 27-d) is live on the hot path; the
xyz embedder's output is computed and discarded by the reference
(if_clight_renderer.py:514-515) and is therefore not evaluated at all here."""
from ..config import cfg_get

view_dim = 3 + 6 * int(cfg_get("view_res", 4))
xyz_dim = 3 + 6 * int(cfg_get("xyz_res", 10))


def view_embedder(viewdir):
    """viewdir [...,3] (already normalised by the caller, like :525-526) -> [...,27]."""
    from .. import hip
    import torch
    sh = viewdir.shape
    # th_view_embed normalises again; a unit vector is a fixed point of d/|d| up to 1 ulp
    return hip.view_embed(viewdir.reshape(-1, 3), int(cfg_get("view_res", 4))).reshape(*sh[:-1], view_dim)
