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


def test_tune_table_roundtrip_and_rejects_foreign_entries(tmp_path):
    """ffn_igemm_tune_export / _import: the tuned (problem -> configuration) table as data.  Entry layout [stamp | key | cfg | splitk]:
    entries with another build's stamp, an unknown configuration or a K split < 1 are skipped (a table written by another build must
    not crash or mis-launch); clear / enable; a tune file that is not a 2-D integer tensor is ignored.  (Whether an
    accepted entry FITS its problem is checked at its first launch, on the GPU: tests/test_ops_gpu.py.)"""
    import torch
    from freefine_amd import _lib, ops
    lib = _lib.load()
    w = lib.ffn_igemm_tune_entry_ints()
    ops.tune_table_clear()
    assert ops.tune_table_export().shape == (0, w)
    t = torch.zeros(4, w, dtype=torch.int32)
    t[:, 0] = lib.ffn_igemm_tune_stamp()
    t[:, 1] = torch.tensor([111, 222, 333, 444]); t[:, 2] = 64; t[:, 3] = 128
    t[:, -2] = torch.tensor([2, lib.ffn_igemm_num_configs() + 5, 0, 2]); t[:, -1] = torch.tensor([1, 1, 0, 1])      # entry 1: unknown cfg; entry 2: split 0
    t[3, 0] ^= 0x100                                                                                                # entry 3: another build's stamp
    assert ops.tune_table_import(t) == 1
    after = ops.tune_table_export()
    assert after.shape[0] == 1 and after[0].tolist() == t[0].tolist()
    f = tmp_path / "tune.pt"
    ops.tune_table_save(str(f))
    assert ops.tune_table_clear() == 1 and ops.tune_table_load(str(f)) == 1
    assert ops.tune_table_load(str(tmp_path / "missing.pt")) == 0
    torch.save({"not": "a table"}, str(tmp_path / "bad.pt"))
    assert ops.tune_table_load(str(tmp_path / "bad.pt")) == 0
    torch.save(torch.zeros(5), str(tmp_path / "bad2.pt"))
    assert ops.tune_table_load(str(tmp_path / "bad2.pt")) == 0
    assert ops.tune_enable(False) is True and ops.tune_enable(True) is False
    ops.tune_table_clear()


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


def test_legacy_vae_attention_keys_are_renamed_and_bad_checkpoints_fail_clearly():
    """SD-1.5 / SD-2.1 VAE safetensors on the hub name the mid-block attention query/key/value/proj_attn (4-D 1x1 conv weights);
    the loader maps them to to_q/to_k/to_v/to_out.0 like diffusers does, and a checkpoint that does not fit the topology is
    rejected with a message naming the missing parameters."""
    import pytest
    import torch
    from freefine_amd.config import VAEConfig
    from freefine_amd.weights import normalize_state_dict, synthetic_state, vae_param_shapes, validate_state_dict
    shapes = vae_param_shapes(VAEConfig.preset("tiny"))
    st = synthetic_state(shapes, 3)
    legacy = {}
    ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    for k, v in st.items():
        for new, old in ren.items():
            if f".attentions.0.{new}." in k:
                k = k.replace(f".{new}.", f".{old}.")
                if v.ndim == 2:
                    v = v[:, :, None, None]
        legacy[k] = v
    assert any(".query.weight" in k for k in legacy) and not any(".to_q." in k for k in legacy)
    fixed = normalize_state_dict(legacy)
    assert set(fixed) == set(st) and all(torch.equal(fixed[k], st[k]) for k in st)
    validate_state_dict(fixed, shapes, "vae")
    broken = dict(fixed)
    broken.pop("decoder.conv_out.bias")
    with pytest.raises(ValueError, match="decoder.conv_out.bias"):
        validate_state_dict(broken, shapes, "vae")


def test_kv64_contract_is_checked_before_any_launch():
    """FFN_IG_OUT_KV64 (round 6): ffn_igemm validates the image-writing epilogue's contract on the host -- wrong arithmetic mode, a boundary that is not a whole
    64-column block, a transposed row stride that is not whole blocks, a residual beside it -- and reports -EINVAL with a message naming the flag (no GPU needed:
    nothing is launched)."""
    import ctypes
    from freefine_amd import _lib
    lib = _lib.load()

    def desc(**kw):
        d = _lib.IgemmDesc()
        d.A = d.W = d.out = 0x10000                   # never dereferenced: validation fails first
        d.M, d.N, d.K, d.Kpad = 4096, 640, 320, 640
        d.lda, d.ldo, d.x3, d.a_lo = 640, 640, 2, 32
        d.rows_per_batch, d.alpha, d.splitk = 4096, 1.0, 1
        d.flags, d.kv64_from = _lib.IG_OUT_KV64, 320
        for k, v in kw.items():
            setattr(d, k, v)
        return d
    for dtype, d in ((_lib.FFN_BF16, desc()),                                           # not split-bf16
                     (_lib.FFN_BF16X3, desc(kv64_from=352)),                             # not a 64-column boundary
                     (_lib.FFN_BF16X3, desc(residual=0x20000, ldr=640)),                 # the image cannot carry a residual
                     (_lib.FFN_BF16X3, desc(flags=_lib.IG_OUT_KV64 | _lib.IG_OUT_TRANSPOSED, N=320, ldo=4096 + 8, kv64_from=0))):      # row stride not whole blocks
        assert lib.ffn_igemm(None, dtype, ctypes.byref(d)) == -22
        assert b"KV64" in lib.ffn_last_error(), lib.ffn_last_error()


def test_kv_images_eligibility_rules():
    """ops.kv_images_ok: which self-attention calls may take K / V^T as the pre-split images their projections write (shapes of the one-wave-per-SIMD kernel, no
    degenerate uniform-softmax entry in the plan, buffers the kernels' 32-bit offsets reach)."""
    from freefine_amd import _lib, ops
    km = object()
    ok = [[ops.AttnEntrySpec(0, 1, 0.0, 1.0, kmask=km), ops.AttnEntrySpec(1, 1, 1.0, -1.0)]]
    uni = [[ops.AttnEntrySpec(0, 0, 1.0, 0.0, kmask=km, flags=_lib.ATT_UNIFORM_SEL0)]]
    skipped = [[ops.AttnEntrySpec(0, 0, 0.0, 0.0, kmask=km, flags=_lib.ATT_UNIFORM_SEL1), None]]        # an inactive term does not count
    assert ops.kv_images_ok(64, 4096, 4096) and ops.kv_images_ok(64, 128, 128, ok) and ops.kv_images_ok(64, 1024, 1024, skipped)
    assert not ops.kv_images_ok(64, 1024, 1024, uni)
    assert not ops.kv_images_ok(40, 4096, 4096) and not ops.kv_images_ok(64, 64, 64) and not ops.kv_images_ok(64, 4096, 77)
    assert not ops.kv_images_ok(64, 4096, 4096, None, 2 ** 31)


def test_attention_row_split_minimises_resident_rounds():
    """ops.attn_row_split: rows per launch of a self-attention call on the one-workgroup-per-CU kernels (256 CUs): never more than FFN_ATT_MAXB, never more rounds
    than the fixed 16-row split, the documented cases, every row covered once."""
    from freefine_amd import _lib, ops
    M = _lib.ATT_MAXB

    def rounds(Bo, w, n, cus=256):
        full, rest = divmod(Bo, n)
        return full * -(-n * w // cus) + (-(-rest * w // cus) if rest else 0)
    assert ops.attn_row_split(72, 80, M, 256) == 16 and rounds(72, 80, 16) == 23
    assert ops.attn_row_split(72, 40, M, 256) == 12 and rounds(72, 40, 12) == 12 and rounds(72, 40, 16) == 14
    assert ops.attn_row_split(72, 20, M, 256) == 12 and rounds(72, 20, 12) == 6 and rounds(72, 20, 16) == 9
    assert ops.attn_row_split(48, 80, M, 256) == 16
    for Bo in (1, 2, 3, 7, 16, 17, 24, 48, 72, 96, 100):
        for w in (5, 20, 40, 80, 180):
            n = ops.attn_row_split(Bo, w, M, 256)
            assert 1 <= n <= min(M, Bo) and rounds(Bo, w, n) <= rounds(Bo, w, min(M, Bo))
