"""CPU: the C-ABI shared library loads and exports every symbol include/freefine_hip.h declares (no compute calls)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from freefine_amd import _lib
    header = open(os.path.join(ROOT, "include", "freefine_hip.h")).read()
    declared = set(re.findall(r"\b(ffn_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    assert lib.ffn_version() >= 1
    for name in declared:
        assert hasattr(lib, name)


def test_invalid_arguments_are_reported_not_crashed():
    import ctypes
    from freefine_amd import _lib
    lib = _lib.load()
    d = _lib.IgemmDesc()
    assert lib.ffn_igemm(None, 0, ctypes.byref(d)) == -22
    assert b"igemm" in lib.ffn_last_error()
    a = _lib.AttnDesc()
    assert lib.ffn_attn(None, 7, ctypes.byref(a)) == -22


def test_controller_plan_tables_host_logic():
    """Attention_Modulator.plan: dispatch + counters mirror the reference protocol (no GPU needed: vectors on CPU)."""
    import torch
    from freefine_amd.attention import Attention_Modulator
    c = Attention_Modulator(start_layer=10)
    c.num_att_layers = 32
    m = torch.zeros(128, 128, dtype=torch.uint8)
    m[30:60, 30:60] = 1
    c.fg_retain_mask = c.fg_ref_mask = c.local_edit_region = m
    c.use_tca, c.method, c.local_edit, c.context_guidance, c.layer_idx = True, "tca", True, 0.5, list(range(10, 16))
    branches = []
    for i in range(32):
        is_cross = bool(i % 2)
        place = "down" if i < 12 else ("mid" if i < 14 else "up")
        branches.append(c.plan("edit", is_cross, place, 4, 64, 2, "cpu")["branch"])
    assert (c.cur_att_layer, c.cur_step) == (0, 1)
    assert branches[0] == "plain" and branches[1] == "cross_local"
    assert [b for b in branches if b.startswith("tca")] == ["tca:tca"] * 6          # blocks 10..15 only
    assert branches[14] == "plain"                                                     # first 'up' self-attn is block 7
