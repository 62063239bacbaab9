"""CPU: the C-ABI library loads and exports every symbol include/scm_hip.h declares;
argument validation (no GPU work) follows the error convention."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "scm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scm_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(g.LIB):
        g.build()
    from scm_gan_amd import _lib
    return _lib.lib()


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libscmgan_hip.so does not export %s" % n
    # and the ctypes binding table covers the same set
    assert set(lib._scm_symbols) == set(names)


def test_version_and_error_convention(lib):
    assert lib.scm_version() >= 100
    from scm_gan_amd import _lib as L
    # null descriptor -> SCM_EINVAL with a message, nothing launched
    assert lib.scm_conv3x3(None, None) == -1
    assert b"null descriptor" in lib.scm_last_error()
    d = L.ConvDesc()
    assert lib.scm_conv3x3(ctypes.byref(d), None) == -1
    assert b"x1" in lib.scm_last_error()
    assert lib.scm_conv3x3_wgrad(None, None) == -1
    assert lib.scm_sn_power_iter(0, None, 1, None) == -1
    assert lib.scm_bernoulli(None, None, None, 0, 0, 0, 0, None, None) == -1
    assert lib.scm_adam_clip_multi(1, None, None, None, None, None, 1e-4, 0.9, 0.999, 1e-8, 1, None, 0.1, 1.0, None) == -1
    assert lib.scm_decoder_fused_fwd(None, None) == -1 and lib.scm_decoder_fused_bwd(None, None) == -1
    assert lib.scm_decoder_fused_pack(None, None, None, None, 16, 3, None, None) == -1
    with pytest.raises(RuntimeError):
        L.check(-1, "scm_conv3x3")


def test_pack_and_workspace_sizes(lib):
    # [groups][cin][9][32][CT]
    assert lib.scm_conv3x3_pack_floats(128, 128) == 128 * 9 * 128
    assert lib.scm_conv3x3_pack_floats(256, 16) == 256 * 9 * 32
    assert lib.scm_conv3x3_pack_floats(64, 48) == 64 * 9 * 64
    assert lib.scm_conv3x3_pack_floats(8, 256) == 2 * 8 * 9 * 128
    pf, cf = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.scm_conv3x3_wgrad_workspace(64, 64, 64, 128, 128, ctypes.byref(pf), ctypes.byref(cf)) == 0
    nsplit = cf.value // 128
    assert pf.value == nsplit * 9 * 128 * 128 and nsplit % 64 == 0
    assert lib.scm_conv2d_bwd_workspace(64, 32, 6, 3) > 0


def test_product_has_no_cpu_fallback():
    """ops refuse CPU tensors instead of silently computing elsewhere."""
    import torch
    from scm_gan_amd import models
    enc = models.Encoder(16, 3) if not torch.cuda.is_available() else None
    if enc is None:
        pytest.skip("GPU present: covered by the gpu tests")
    with pytest.raises(RuntimeError):
        enc(torch.zeros(1, 3, 3, 8, 8))
