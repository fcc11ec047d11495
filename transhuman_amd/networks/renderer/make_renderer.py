"""make_renderer -- /root/reference/lib/networks/renderer/make_renderer.py:4-8.

The reference loads cfg.renderer_path with imp.load_source; the drop-in is to
point `renderer_module`/`renderer_path` in the YAML at this package's files
(INTEGRATION.md).  Standalone use: make_renderer(cfg, network)."""
import importlib


def make_renderer(cfg, network, **kw):
    module = getattr(cfg, "renderer_module", "transhuman_amd.networks.renderer.if_clight_renderer")
    if module.startswith("lib.networks.renderer."):
        module = "transhuman_amd.networks.renderer." + module.rsplit(".", 1)[1]
    return importlib.import_module(module).Renderer(network, **kw)
