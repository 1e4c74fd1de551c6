"""CPU: the C-ABI library loads and exports every symbol include/swn_hip.h declares
(no compute calls without a GPU), and the host-side packer honours the documented layout."""
import ctypes
import os
import re

import numpy as np
import pytest

from shallow_wavenet_amd import _lib, config as C
from shallow_wavenet_amd.runtime import pack_state_dict
from shallow_wavenet_amd.synth import synth_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "swn_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(swn_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib.SIGNATURES, f"{n} declared in the header but not bound in _lib.py"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.swn_abi_version() == 3
    assert lib.swn_strerror(-1).decode().startswith("network descriptor")


@pytest.mark.parametrize("cfg", [C.bl6_laplace(), C.bl6_laplace(5, 4), C.bl6_softmax(), C.ref6_laplace(),
                                 C.ref6_softmax(), C.tiny(), C.tiny("softmax", wav_conv_flag=True)])
def test_geometry_agrees_with_host_config(cfg):
    lib = _lib.lib()
    d = _lib.desc_from_cfg(cfg)
    assert lib.swn_receptive_field(ctypes.byref(d)) == cfg.receptive_field
    assert lib.swn_num_tensors(ctypes.byref(d)) == len(cfg.param_shapes())
    assert lib.swn_packed_floats(ctypes.byref(d)) > cfg.n_params() * 0.9


def test_precision_is_a_call_argument_not_library_state():
    """ABI 3: no process-wide arithmetic switch is exported; the training entry points take SWN_PRECISION_* and reject
    anything else before touching a device (no GPU call behind this test)."""
    lib = _lib.lib()
    assert not hasattr(lib, "swn_train_set_precision") and not hasattr(lib, "swn_train_get_precision")
    d = _lib.desc_from_cfg(C.bl6_laplace())
    r = ctypes.byref(d)
    null = ctypes.c_void_p(None)
    assert lib.swn_backward(r, null, null, null, null, null, null, null, null, 1, 1, null, null, 7, null) == -2
    assert lib.swn_backward_drop(r, null, null, null, null, null, null, null, null, null, 1, 1, null, null, 2, null) == -2
    assert lib.swn_forward_drop(r, null, null, null, 1, 1, null, null, null, null, null, -1, null) == -2
    assert lib.swn_bf16_work_to_f32(r, null, null, 1, 1, null, 3, null) == -2
    from shallow_wavenet_amd import runtime
    assert runtime.current_precision() == _lib.PRECISION_FP32
    with runtime.train_precision("bf16"):
        assert runtime.current_precision() == _lib.PRECISION_BF16
        with runtime.train_precision("fp32"):
            assert runtime.current_precision() == _lib.PRECISION_FP32
        assert runtime.current_precision() == _lib.PRECISION_BF16
    assert runtime.current_precision() == _lib.PRECISION_FP32
    with pytest.raises(ValueError):
        runtime.train_precision("fp8")


def test_bad_descriptors_are_rejected():
    lib = _lib.lib()
    d = _lib.desc_from_cfg(C.bl6_laplace())
    d.kernel_size = 1
    assert lib.swn_receptive_field(ctypes.byref(d)) == -1
    d = _lib.desc_from_cfg(C.bl6_laplace(5, 4))
    d.seg = 11
    assert lib.swn_packed_floats(ctypes.byref(d)) == 0
    # the (seg,1) Conv2d is folded into in_x: two more tensors, same packed size
    d = _lib.desc_from_cfg(C.bl6_laplace(5, 4))
    n0, p0 = lib.swn_num_tensors(ctypes.byref(d)), lib.swn_packed_floats(ctypes.byref(d))
    d.aux_conv2d_flag = 1
    assert lib.swn_num_tensors(ctypes.byref(d)) == n0 + 2
    assert lib.swn_packed_floats(ctypes.byref(d)) == p0
    with pytest.raises(KeyError):
        pack_state_dict(C.tiny(), {})


def test_pack_layout_known_answers():
    """spot-check the packed layout against the formulas documented in csrc/swn_geom.hpp."""
    cfg = C.tiny("laplace", 2, 4)
    sd = synth_state_dict(cfg, seed=3)
    packed = pack_state_dict(cfg, sd).numpy()
    H, K, L, seg = cfg.H, cfg.K, cfg.L, cfg.seg

    def al(x):
        return (x + 63) & ~63

    o = 0
    o_scale_w = o
    o = al(o + cfg.n_aux ** 2)
    o = al(o + cfg.n_aux)
    for i in range(cfg.aux_dilation_size):
        cin, cout = cfg.n_aux * 3 ** i, cfg.n_aux * 3 ** (i + 1)
        o = al(o + cout * cin * 3)
        o = al(o + cout)
    o_wx = o
    A0p = (cfg.A0 + 3) & ~3
    assert np.array_equal(packed[o_scale_w:o_scale_w + cfg.n_aux ** 2], sd["scale_in.weight"].ravel())
    # wx row n=(l*seg+s)*2H+o holds in_x[l].weight[o, c*seg+s]
    l, s, oo, c = 3, 1, 17, 40
    n = (l * seg + s) * 2 * H + oo
    assert packed[o_wx + n * A0p + c] == sd[f"in_x.{l}.weight"][oo, c * seg + s, 0]
    o = al(o_wx + L * seg * 2 * H * A0p)
    o_wup = o
    o = al(o + cfg.U)
    o = al(o + 1)
    assert np.array_equal(packed[o_wup:o_wup + cfg.U], sd["upsampling.conv.weight"].ravel())
    o_bx = o
    o = al(o + L * 2 * H)
    want = sd["in_x.2.bias"][5] + sd["upsampling.conv.bias"][0] * sd["in_x.2.weight"][5].astype(np.float64).sum()
    assert abs(packed[o_bx + 2 * 2 * H + 5] - want) < 1e-6
    o_bxr = o                                  # raw in_x bias (dropout mode evaluates in_x at sample rate)
    o = al(o + L * 2 * H)
    assert packed[o_bxr + 2 * 2 * H + 5] == sd["in_x.2.bias"][5]
    o = al(o + H)
    o_cv = o
    o = al(o + K * H)
    o_cc = o
    o = al(o + K * H)
    wc, ww, wb = sd["causal.conv.weight"], sd["wav_conv.weight"][:, 0, 0], sd["wav_conv.bias"]
    assert abs(packed[o_cv + 1 * H + 7] - (wc[7, :, 1].astype(np.float64) @ ww)) < 1e-6
    assert abs(packed[o_cc + 2 * H + 9] - (wc[9, :, 2].astype(np.float64) @ wb)) < 1e-6
    o_wd = o
    Hp = (H + 3) & ~3
    assert packed[o_wd + ((4 * 2 * H + 33) * K + 2) * Hp + 11] == sd["dil_h.4.conv.weight"][33, 11, 2]


def test_in_tree_library_was_built_from_the_sources_in_the_tree():
    """build() rebuilds when the recorded source hash (csrc/*, Makefile, include/swn_hip.h) differs from the tree:
    a stale .so cannot pass silently."""
    from shallow_wavenet_amd import _lib
    _lib.lib()
    assert os.path.exists(_lib.STAMP_PATH), "library without a source-hash stamp: run __graft_entry__.build()"
    assert not _lib.is_stale(), "libswn_hip.so does not match csrc/: run __graft_entry__.build()"
    h = _lib.source_hash()
    assert len(h) == 64 and h == _lib.source_hash()
