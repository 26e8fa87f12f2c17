"""CPU: host logic of the reference-stream reuse and of the image-batched pass tables (no kernel is launched: HipUNet only packs
weights in its constructor, which is plain tensor plumbing and runs on the CPU device)."""
import torch

from freefine_amd import ops
from freefine_amd.attention import Attention_Modulator
from freefine_amd.config import UNetConfig
from freefine_amd.unet import HipUNet
from freefine_amd.weights import synthetic_state, unet_param_shapes


def _net():
    cfg = UNetConfig.preset("tiny")
    return HipUNet(cfg, synthetic_state(unet_param_shapes(cfg), 0), dtype=torch.float32, device="cpu")


def _controller(net, hook="edit"):
    c = Attention_Modulator(start_layer=10)
    net.set_attention_control(hook, c)
    c.num_att_layers = net.num_attention_calls
    m = torch.zeros(128, 128, dtype=torch.uint8)
    m[30:60, 30:60] = 1
    c.fg_retain_mask = c.fg_ref_mask = c.local_edit_region = m
    c.use_tca, c.method, c.local_edit, c.context_guidance, c.layer_idx = True, "tca", True, 0.5, list(range(10, 16))
    return c


def test_join_block_is_the_up_block_of_the_first_modulated_layer():
    net = _net()
    assert net.num_attention_calls == 32
    # transformer blocks: down 0-5, mid 6, up_blocks[1] 7-9, up_blocks[2] 10-12, up_blocks[3] 13-15
    assert net.join_block(10) == (2, 10) and net.join_block_tb(2) == 10
    assert net.join_block(12) == (2, 10)          # a first modulated block inside up_blocks[2] still joins at its entrance
    assert net.join_block(13) == (3, 13) and net.join_block(7) == (1, 7)
    assert net.join_block(None) == (3, 13)        # nothing modulated: join as late as possible
    assert net.join_block(3) == (0, 7) or net.join_block(3)[0] == 0      # modulation before the up path: no block may be skipped


def test_replay_index_maps_and_phase_tables():
    """physical rows per image (edit_u, ref, edit_c): phase A holds rows 0 and 2 of every image; the merged batch restores physical order;
    the cross-attention table of phase A is renumbered to its two rows, the TCA table of phase B spans all three and keeps the tiled-head
    rule pinned to the LOGICAL row."""
    net = _net()
    c = _controller(net)
    net._row_map = (0, 1, 2, 1)
    enc = torch.randn(3, 77, 64)
    ru = net._prepare_reuse(dict(mode="replay", join=2, ref=(False, True, False), state=[torch.zeros(1, 1, 1)]), 3, enc)
    assert ru["sel"] == [0, 2] and ru["idx_a"].tolist() == [0, 2] and ru["perm"].tolist() == [0, 2, 1]
    assert torch.equal(ru["enc_a"], enc[[0, 2]])
    net._reuse = ru
    net._in_phase_a = True
    c.cur_att_layer = 1                                     # a cross-attention call of block 0
    plan = net._plan(True, "down", 2, 256, 2)
    assert plan["branch"] == "cross_local" and [len(r) for r in plan["passes"]] == [2, 2]
    p0, p1 = plan["passes"]
    assert (p0[0].q_row, p0[0].kv_row, p0[1].q_row, p0[1].kv_row) == (0, 0, 1, 1) and p0[1].wq is not None
    assert p1[0] is None and (p1[1].q_row, p1[1].kv_row) == (0, 0)          # c_e blends with u_e = phase-A row 0
    assert p0[1].hr_row == 2                                               # logical row of physical row 2
    net._in_phase_a = False
    c.cur_att_layer = 20                                    # self-attention of block 10: TCA, all three physical rows
    plan = net._plan(False, "up", 3, 256, 2)
    assert plan["branch"] == "tca:tca" and [len(r) for r in plan["passes"]] == [3, 3]
    ref_pass = plan["passes"][0]
    assert [(e.q_row, e.kv_row) for e in ref_pass] == [(0, 1), (1, 1), (2, 1)] and [e.hr_row for e in ref_pass] == [0, 1, 2]
    # two images, two physical rows each (empty edit prompt: cond == uncond): image-major maps
    net2 = _net()
    cs = [_controller(net2), None]
    cs[1] = Attention_Modulator(start_layer=10)
    for k in ("num_att_layers", "fg_retain_mask", "fg_ref_mask", "local_edit_region", "use_tca", "method", "local_edit", "context_guidance", "layer_idx"):
        setattr(cs[1], k, getattr(cs[0], k))
    net2.set_attention_control("edit", cs)
    net2._row_map = (0, 1, 0, 1)
    ru = net2._prepare_reuse(dict(mode="replay", join=2, ref=(False, True), state=[torch.zeros(2, 1, 1)]), 4, torch.randn(4, 77, 64))
    assert ru["idx_a"].tolist() == [0, 2] and ru["perm"].tolist() == [0, 2, 1, 3]
    # no de-duplication: four physical rows, two of them reference rows fed from ONE recorded row (repeat_interleave on the caller's side)
    net2._row_map = None
    ru = net2._prepare_reuse(dict(mode="replay", join=2, ref=(False, True, False, True), state=[torch.zeros(4, 1, 1)]), 8, torch.randn(8, 77, 64))
    assert ru["idx_a"].tolist() == [0, 2, 4, 6] and ru["perm"].tolist() == [0, 4, 1, 5, 2, 6, 3, 7]


def test_dropped_tail_keeps_reference_keys_but_not_reference_queries():
    """every guided step but the last: the reference rows stop after the K / V projection of the last transformer block -- its self
    attention keeps reading their K / V (physical rows) but has no output row for them, the tiled-head rule stays pinned"""
    net = _net()
    c = _controller(net)
    net._row_map = (0, 1, 2, 1)
    ru = net._prepare_reuse(dict(mode="replay", join=2, ref=(False, True, False), state=[torch.zeros(1, 1, 1)], drop_tail=True), 3, torch.randn(3, 77, 64))
    assert ru["idx_a_list"] == [0, 2]
    net._reuse = ru
    c.cur_att_layer = 30                                    # self-attention of block 15
    plan = net._without_ref_queries(net._plan(False, "up", 3, 256, 2), 3)
    assert [len(r) for r in plan["passes"]] == [2, 2]
    assert [(e.q_row, e.kv_row, e.hr_row) for e in plan["passes"][0]] == [(0, 1, 0), (2, 1, 2)]
    assert [(e.q_row, e.kv_row) for e in plan["passes"][1]] == [(0, 0), (2, 2)]
    # a plain (unmodulated) call: explicit one-pass table over the surviving rows, head rule pinned to the physical row
    plain = net._without_ref_queries(dict(kind="passes", passes=None, needs_cg=False), 3)
    assert [(e.q_row, e.kv_row, e.hr_row) for e in plain["passes"][0]] == [(0, 0, 0), (2, 2, 2)]
    # the fingerprint of a dropped-tail forward differs from the full one's in exactly the last two attention calls
    fps = {}
    for drop in (False, True):
        ru["drop_tail"] = drop
        c.cur_att_layer, c.cur_step = 0, 0
        fps[drop] = net._plan_all_slow(3, 16, 16)[1]
    assert fps[False][:-2] == fps[True][:-2] and fps[False][-2] != fps[True][-2] and fps[False][-1] != fps[True][-1]
    assert len(fps[True][-2][2][0]) == 2 and len(fps[True][-1][2][0]) == 2 and len(fps[False][-2][2][0]) == 3


def test_phase_a_rejects_terms_that_reach_a_reference_row():
    e = ops.AttnEntrySpec(0, 1)
    try:
        e.renumber({0: 0, 2: 1})
    except ValueError as err:
        assert "outside the rows of this phase" in str(err)
    else:
        raise AssertionError("a phase-A term must not read a reference row")


def test_batched_composition_shifts_text_rows_by_the_text_block():
    """composition hook, K = 2 images, R = 2 references, P = 3 prompts: 4 latent rows and 6 text rows per image -- the cross-attention
    K / V rows of image 1 start at text row 6, its query rows at latent row 4"""
    net = _net()
    cs = []
    for _ in range(2):
        c = Attention_Modulator(start_layer=10)
        c.num_att_layers = net.num_attention_calls
        m = torch.zeros(128, 128, dtype=torch.uint8)
        m[30:60, 30:60] = 1
        c.src_masks, c.tgt_masks = torch.stack([m, m]), torch.stack([m, m, 1 - m])
        c.use_tca, c.method, c.local_edit, c.context_guidance, c.layer_idx, c.prompt_length = True, "tca", True, 0.5, list(range(10, 16)), 3
        cs.append(c)
    net.set_attention_control("compose", cs)
    net._row_map, net._enc_rows = None, 12
    for c in cs:
        c.cur_att_layer = 1
    plan = net._plan(True, "down", 8, 256, 2)
    assert len(plan["passes"]) == 3 and all(len(r) == 8 for r in plan["passes"])
    last0, last1 = [r[3] for r in plan["passes"]], [r[7] for r in plan["passes"]]
    assert [(e.q_row, e.kv_row) for e in last0] == [(3, 3), (3, 4), (3, 5)]
    assert [(e.q_row, e.kv_row) for e in last1] == [(7, 9), (7, 10), (7, 11)]
    assert (plan["passes"][0][5].q_row, plan["passes"][0][5].kv_row) == (5, 7)      # reference row 1 of image 1: latent row 5, text row 6 + 1


def test_stored_kv_composition_tables():
    """composition hook, stored reference K / V (replay_kv): K = 2 images, R = 2 references, P = 3 prompts.  The launch holds the two edit
    rows of every image; the self attention of a recorded block reads the references' K / V from the rows BEHIND the 2 K computed ones
    (image-major), the cross attention reads the 1 + P surviving text rows of its image."""
    net = _net()
    cs = []
    for _ in range(2):
        c = Attention_Modulator(start_layer=10)
        c.num_att_layers = net.num_attention_calls
        m = torch.zeros(128, 128, dtype=torch.uint8)
        m[30:60, 30:60] = 1
        c.src_masks, c.tgt_masks = torch.stack([m, m]), torch.stack([m, m, 1 - m])
        c.use_tca, c.method, c.local_edit, c.context_guidance, c.layer_idx, c.prompt_length = True, "tca", True, 0.5, list(range(10, 16)), 3
        cs.append(c)
    net.set_attention_control("compose", cs)
    net._row_map, net._enc_rows = None, 12
    R, P, kv_from = 2, 3, 10
    kv = [(torch.zeros(4, 1, 1), torch.zeros(4, 1, 1)) for _ in range(len(net.transformers) - kv_from)]
    enc = torch.randn(12, 77, 64)
    ru = net._prepare_reuse(dict(mode="replay_kv", kv_from=kv_from, ref=(False, True, True, False), kv=kv, text_sel=[0, 3, 4, 5]), 8, enc)
    assert ru["sel"] == [0, 3] and ru["refs"] == [1, 2] and ru["idx_a"].tolist() == [0, 3, 4, 7] and len(ru["state"]) == 12
    assert torch.equal(ru["enc_a"], enc[[0, 3, 4, 5, 6, 9, 10, 11]])
    net._reuse = ru
    for c in cs:
        c.cur_att_layer = 1                                 # cross attention of block 0: edit_c blends the P prompts, per image
    plan = net._plan(True, "down", 4, 256, 2)
    assert len(plan["passes"]) == 3 and all(len(r) == 4 for r in plan["passes"])
    assert [(e.q_row, e.kv_row) for e in plan["passes"][0]] == [(0, 0), (1, 1), (2, 4), (3, 5)]
    assert [None if e is None else (e.q_row, e.kv_row) for e in plan["passes"][1]] == [None, (1, 2), None, (3, 6)]
    assert [None if e is None else (e.q_row, e.kv_row) for e in plan["passes"][2]] == [None, (1, 3), None, (3, 7)]
    for c in cs:
        c.cur_att_layer = 20                                # self attention of block 10: K / V rows [e0_u, e0_c, e1_u, e1_c | r0_1, r0_2, r1_1, r1_2]
    net._kv_ext = True
    plan = net._plan(False, "up", 4, 256, 2)
    net._kv_ext = False
    assert [(e.q_row, e.kv_row) for e in plan["passes"][0]] == [(0, 0), (1, 1), (2, 2), (3, 3)]          # the edit rows' own K / V
    assert [(e.q_row, e.kv_row) for e in plan["passes"][1]] == [(0, 4), (1, 4), (2, 6), (3, 6)]          # reference 1 of the own image
    assert [(e.q_row, e.kv_row) for e in plan["passes"][2]] == [(0, 5), (1, 5), (2, 7), (3, 7)]          # reference 2
    # the whole-forward fingerprint marks exactly the self-attention calls of blocks >= kv_from as extended
    for c in cs:
        c.cur_att_layer, c.cur_step = 0, 0
    fp = net._plan_all_slow(8, 16, 16)[1]
    ext = [i for i, f in enumerate(fp) if f != 0 and any(e is not None and e[1] >= 4 for r in f[2] for e in r) and i % 2 == 0]
    assert ext == [2 * t for t in range(10, 16)]
