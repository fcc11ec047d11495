"""CPU: the C-ABI shared library loads (no GPU needed) and exports every symbol
include/transhuman_hip.h declares; the ctypes binding table matches the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "transhuman_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(th_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    from transhuman_amd import build, hip
    build.build(force=False, verbose=False)
    return hip.load_library()


def test_header_declares_the_expected_surface():
    names = header_functions()
    for must in ("th_ctx_create", "th_ctx_destroy", "th_last_error", "th_set_mlp_weights", "th_set_vit_weights",
                 "th_hull_mask", "th_paint_group", "th_vit_forward", "th_dparf_encode", "th_pixel_gather",
                 "th_network_forward", "th_composite", "th_render_rays", "th_eval_sigma_grid"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    raw = ctypes.CDLL(os.path.join(ROOT, "transhuman_amd", "libtranshuman_hip.so"))
    missing = [n for n in header_functions() if not hasattr(raw, n)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_binding_table_matches_header(lib):
    from transhuman_amd import hip
    assert sorted(hip.SYMBOLS) == header_functions()
    assert lib.th_abi_version() == 12


def test_workspace_queries_are_pure_host_calls(lib):
    # no device needed for sizing
    assert lib.th_linear_workspace_bytes(256, 255) >= 256 * 256 * 4
    assert lib.th_hull_workspace_bytes(6890) > 6890 * 12
    assert lib.th_vit_workspace_bytes(3, 500, 192, 3) >= 3 * 500 * 192 * 4 * 6
    assert lib.th_network_workspace_bytes(3, 1000) > 3 * 1000 * (256 + 384) * 4


def test_no_fallback_without_a_device(lib):
    """the product path must fail loudly when there is no GPU / library, never compute on the CPU"""
    import torch
    from transhuman_amd import hip
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hip.HipError):
        hip.ctx()
    from util import make_net
    net = make_net(2)
    with pytest.raises((hip.HipError, AssertionError)):
        net.ViT(torch.zeros(1, 8, 192), torch.zeros(1, 8, 3))


def test_errors_are_reported_not_crashed(lib):
    bad = lib.th_set_chunk_samples(3)
    assert bad != 0 and b"chunk" in lib.th_last_error()
    assert lib.th_set_chunk_samples(524288) == 0


def _integration_snippet_structs():
    """The ctypes Structure classes of INTEGRATION.md's python stub, executed as written."""
    import ctypes as C
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    assert blocks, "INTEGRATION.md lost its ctypes stub"
    out = {}
    for blk in blocks:
        for m in re.finditer(r"^class (\w+)\(C\.Structure\):.*?\n(?:[ \t]+.*\n)+", blk, flags=re.M):
            ns = {"C": C}
            exec(m.group(0), ns)                      # (documentation of this repository, not reference content)
            out[m.group(1)] = ns[m.group(1)]
    return out


def test_integration_md_structs_match_the_library(lib):
    """the document's ThPoints once had 9 fields against the ABI's 11 -- a maintainer who copied it
    handed th_composite a struct 16 bytes short.  The snippet is now executed and held to th_sizeof()."""
    import ctypes as C
    from transhuman_amd import hip
    structs = _integration_snippet_structs()
    assert "ThPoints" in structs
    cname = {"ThPoints": b"th_points", "ThFrame": b"th_frame", "ThMapSource": b"th_map_source"}
    for name, cls in structs.items():
        want = lib.th_sizeof(cname[name])
        assert want > 0 and C.sizeof(cls) == want, (name, C.sizeof(cls), want)
        ours = getattr(hip, name)
        assert [f[0] for f in cls._fields_] == [f[0] for f in ours._fields_], name
    # the binding's own structs against the library as compiled
    assert C.sizeof(hip.ThPoints) == lib.th_sizeof(b"th_points")
    assert C.sizeof(hip.ThFrame) == lib.th_sizeof(b"th_frame")
    assert C.sizeof(hip.ThMapSource) == lib.th_sizeof(b"th_map_source")
    assert lib.th_sizeof(b"no_such_struct") == 0
