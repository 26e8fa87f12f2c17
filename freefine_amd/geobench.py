"""GeoBench-2D batch-inference harness on the MI355X engine: counterpart of the reference's driver
/root/reference/evaluation/FreeFine/freefine_batch_infer_2d.py (case list :91-132, model setup :148-157, per-case
pre-processing + FreeFine_generation call :175-236, result gather + JSON :243-262).

Differences from the reference driver, all outside the arithmetic of an edit:
  * cases are sharded with dist.shard_indices (DistributedSampler semantics) and the per-rank results gathered with
    all_gather_object -- as the reference does -- but the process group is optional (single process when WORLD_SIZE is unset);
  * `batch` cases are edited together through FreeFine_generation_batch (the reference must use batch size 1, :170);
  * host pre-processing (PNG decode, resize, the affine coarse edit) runs in a prefetch thread so it overlaps the GPU;
  * the reference reads `ori_mask` after using it in re_edit_2d (:196 vs :198, a NameError on the first case); here it is read first.
"""
import os
import os.path as osp
import queue
import threading

import numpy as np

from src.utils.vis_utils import load_json, re_edit_2d, read_and_resize_img, read_and_resize_mask, save_img, save_json

# the GeoBench-2D call parameters (freefine_batch_infer_2d.py:212-230)
GEOBENCH_2D = dict(guidance_scale=7.5, eta=1.0, end_scale=0.0, end_step=50, num_step=50, start_step=35, seed=42)
GEN_SUBDIR = "Geo-Bench-2D/Gen_results_FreeFine_2d"
INP_SUBDIR = "Geo-Bench-2D/inp_img_blended"
# the GeoBench-3D (depth-guided coarse edit) call parameters (freefine_batch_infer_3d_depth.py:144-162): the coarse input was
# rendered beforehand by the 3-D front end (DepthAnything + point-cloud warp, out of scope here) and is read from disk
GEOBENCH_3D_DEPTH = dict(guidance_scale=7.5, eta=1.0, end_scale=0.0, end_step=50, num_step=50, start_step=15, seed=42)
VARIANTS = {
    "2d": dict(params=GEOBENCH_2D, annotations="annotations_2d.json", gen_subdir=GEN_SUBDIR, results="generated_results_freefine_2d.json"),
    "3d_depth": dict(params=GEOBENCH_3D_DEPTH, annotations="annotations.json", gen_subdir="Geo-Bench-3D/Gen_results_FreeFine_depth",
                     results="generated_results_freefine_depth.json"),
    # the same edit with the coarse input rendered here from the RGB image + transform (depth network + point-cloud warp on the GPU)
    "3d_rgb": dict(params=GEOBENCH_3D_DEPTH, annotations="annotations.json", gen_subdir="Geo-Bench-3D/Gen_results_FreeFine_depth_rgb",
                   results="generated_results_freefine_depth_rgb.json"),
}


class CaseList:
    """CustomDataset of the reference (:91-132): one case per data[da_n]['instances'][ins_id][edit_ins]; cases whose output PNG
    exists are kept aside as existing results."""

    def __init__(self, data, dst_dir_path_gen, check_exist=True):
        self.cases, self.existing_results = [], []
        self.dst = dst_dir_path_gen
        for da_n, da in data.items():
            for ins_id, cur in da.get("instances", {}).items():
                for edit_ins, pack in cur.items():
                    item = dict(da_n=da_n, ins_id=ins_id, edit_ins=edit_ins, **pack)
                    path = self.expected_path(da_n, ins_id, edit_ins)
                    if check_exist and osp.exists(path):
                        item["gen_img_path"] = path
                        self.existing_results.append(item)
                    else:
                        self.cases.append(item)

    def expected_path(self, da_n, ins_id, edit_ins):
        return osp.join(self.dst, str(da_n), str(ins_id), f"{edit_ins}.png")

    def __len__(self):
        return len(self.cases)

    def __getitem__(self, i):
        return self.cases[i]


def load_case(case, dst_base, dsize=(512, 512)):
    """host pre-processing of one case (:187-210): inputs of FreeFine_generation"""
    inp_bg = read_and_resize_img(osp.join(dst_base, INP_SUBDIR, str(case["da_n"]), str(case["ins_id"]), "inp_img.png"), dsize)
    ori_img = read_and_resize_img(case["ori_img_path"], dsize)
    ori_mask = read_and_resize_mask(case["ori_mask_path"], dsize)
    coarse, target_mask = re_edit_2d(ori_img, ori_mask, case["edit_param"], inp_bg)[:2]
    return dict(ori_img=ori_img, ori_mask=ori_mask, coarse_input=coarse, target_mask=target_mask, guidance_text="",
                draw_mask=np.ones_like(ori_mask), use_auto_draw=True, reduce_inp_artifacts=True, cons_area=target_mask)


def load_case_3d_depth(case, dst_base, dsize=(512, 512)):
    """host pre-processing of one GeoBench-3D case (freefine_batch_infer_3d_depth.py:127-143)"""
    coarse = read_and_resize_img(osp.join(dst_base, "coarse3d_depth_anything", str(case["da_n"]), str(case["ins_id"]), f'{case["edit_ins"]}.png'), dsize)
    target_mask = read_and_resize_mask(case["target_mask_0"], dsize)
    return dict(ori_img=read_and_resize_img(case["ori_img_path"], dsize), ori_mask=read_and_resize_mask(case["ori_mask_path"], dsize),
                coarse_input=coarse, target_mask=target_mask, guidance_text=case["obj_label"],
                draw_mask=read_and_resize_mask(case["draw_mask"], dsize), use_auto_draw=False, reduce_inp_artifacts=True, cons_area=target_mask)


def monocular_depth(img, depth_model, translate_factor=0.0, side=518):
    """get_monocular_depth_anything as src/utils/ui_utils.py:380-402 calls it (`get_depth(..., translate_factor=0.0)`; the copy in
    evaluation/DiffusionHandles/eval_geobench.py:163-215 has its push-back line commented out, :214):
    shorter side -> `side` (a multiple of 14, aspect kept, both sides multiples of 14), ImageNet normalisation, the depth network, bilinear back
    to the image size, `depth.max() - depth` (relative -> absolute), pushed back by translate_factor * max (0 by default, like get_depth).
    The resampling around the network is torch (plumbing); the network is freefine_amd.depth.HipDepthAnything.  Runs on the CALLER's thread
    and current device (geobench.run calls it on the consumer thread, never on the prefetch thread)."""
    import torch
    import torch.nn.functional as F
    h, w = img.shape[:2]
    sc = side / min(h, w)
    nh, nw = max(14, int(round(h * sc / 14)) * 14), max(14, int(round(w * sc / 14)) * 14)
    dev = depth_model.device if hasattr(depth_model, "device") else "cuda:0"
    x = torch.from_numpy(np.ascontiguousarray(img)).to(dev).permute(2, 0, 1)[None].float() / 255.0
    x = F.interpolate(x, size=(nh, nw), mode="bicubic", align_corners=False)
    mean = torch.tensor([0.485, 0.456, 0.406], device=x.device)[None, :, None, None]
    std = torch.tensor([0.229, 0.224, 0.225], device=x.device)[None, :, None, None]
    d = depth_model(((x - mean) / std).contiguous()).float()
    d = F.interpolate(d[None] if d.ndim == 3 else d, (h, w), mode="bilinear", align_corners=False)[0, 0]
    d = d.max() - d
    d = d + d.max() * translate_factor
    return d.clamp_min(0).cpu().numpy().astype(np.float32)


def read_case_3d_rgb(case, dst_base, dsize=(512, 512)):
    """HOST half of the 3d_rgb loader (files only: safe on the prefetch thread): image, mask, inpainted background, edit parameters"""
    return dict(_raw_3d_rgb=True, ori_img=read_and_resize_img(case["ori_img_path"], dsize), ori_mask=read_and_resize_mask(case["ori_mask_path"], dsize),
                bg=read_and_resize_img(osp.join(dst_base, INP_SUBDIR, str(case["da_n"]), str(case["ins_id"]), "inp_img.png"), dsize),
                edit_param=[float(v) for v in case["edit_param"]], obj_label=case.get("obj_label", ""), dsize=tuple(dsize))


def finish_case_3d_rgb(raw, depth_model, focal_length=550.0):
    """DEVICE half of the 3d_rgb loader (depth network + point-cloud warp on the HIP kernels): call it on the thread that owns the device
    and the stream -- geobench.run does so on its consumer thread, between batches, never concurrently with a graph capture."""
    from . import warp3d
    assert depth_model is not None, "the 3d_rgb variant needs a depth model (freefine_amd.depth.HipDepthAnything or depth_anything.dpt.DepthAnything)"
    ori_img, ori_mask, bg, dsize = raw["ori_img"], raw["ori_mask"], raw["bg"], raw["dsize"]
    depth = monocular_depth(ori_img, depth_model, translate_factor=0.1)      # this variant's own choice: no point of the cloud at z = 0
    ep = raw["edit_param"]
    sc = dsize[0] / 512.0                                    # edit_param translations are pixels at the 512 reference size
    tf = [ep[0] * sc, ep[1] * sc, ep[2] * sc, ep[3], ep[4], ep[5], ep[6], ep[7], ep[8]]
    m2 = ori_mask if ori_mask.ndim == 2 else ori_mask[:, :, 0]
    coarse, target_mask = warp3d.coarse_edit_3d(ori_img, m2, depth, tf, bg, focal_length=focal_length * dsize[0] / 512.0)
    draw = ndimage_max(np.maximum(target_mask, (m2 > 0).astype(np.uint8) * 255), 9)
    return dict(ori_img=ori_img, ori_mask=ori_mask, coarse_input=coarse, target_mask=target_mask, guidance_text=raw["obj_label"],
                draw_mask=(draw > 0).astype(np.uint8), use_auto_draw=False, reduce_inp_artifacts=True,
                cons_area=np.maximum(target_mask, (m2 > 0).astype(np.uint8) * 255))


def load_case_3d_rgb(case, dst_base, dsize=(512, 512), depth_model=None, focal_length=550.0):
    """a GeoBench-3D case whose coarse edit is built HERE from the RGB image and a 3-D transform instead of being read from disk
    (freefine_batch_infer_3d_depth.py:121 reads coarse3d_depth_anything/...png).  A NON-PARITY EXTENSION, not a restatement of the dataset's
    generator: evaluation/FreeFine/get_3d_transform_correspondence.py:217-251 renders those files through the GeoDiffuser warp
    (`get_transformed_mask`: translation edit_param / 512 in world units composed as T.S.Rx.Ry.Rz, splatting radius 1.3, 15 points per
    pixel, depth from get_depth(translate_factor=0.0)), whose modules are not in the reference tree; here the cloud goes through
    freefine_amd.warp3d (the pytorch3d splat of geo_utils.py:427-528: radius 1.5 px, K = 5), the translation is read as PIXELS at the 512
    reference size at the object's mean depth (warp3d pixel_translation) and the depth is pushed back by 0.1 max.  Coarse images and target
    masks therefore differ systematically from the dataset's coarse3d_depth_anything files; use --variant 3d_depth to reproduce those.
    edit_param = [tx, ty, tz, rx, ry, rz (degrees), sx, sy, sz].  = read_case_3d_rgb (host) + finish_case_3d_rgb (device)."""
    return finish_case_3d_rgb(read_case_3d_rgb(case, dst_base, dsize), depth_model, focal_length)


def _prefetch(cases, dst_base, depth, dsize, loader=None):
    q = queue.Queue(maxsize=depth)

    def work():
        for c in cases:
            try:
                q.put((c, (loader or load_case)(c, dst_base, dsize), None))
            except Exception as e:  # noqa: BLE001 -- reported on the consumer side with the case attached
                q.put((c, None, e))
        q.put(None)

    threading.Thread(target=work, daemon=True).start()
    while True:
        item = q.get()
        if item is None:
            return
        yield item


def warm_and_sync(model, batch, params, dsize=(512, 512), variant="2d", src=0):
    """Multi-rank runs in bf16 fast mode: every rank edits one SYNTHETIC batch of the run's shape eagerly (this tunes every igemm
    shape of the schedule on its own GPU), then adopts rank `src`'s tuning table and stops tuning (dist.sync_tune_table) BEFORE any
    forward graph is captured -- every rank then launches identical tile / split-K configurations, so a case's bf16 result does not
    depend on the rank that edited it.  A collective: every rank calls it (run() does, also on ranks that own no case)."""
    import torch
    from . import dist as FD
    if not FD.active():
        return 0
    unet = getattr(model, "unet", None)
    if unet is not None and getattr(unet, "dtype", None) == torch.bfloat16:      # f32 parity mode never tunes: nothing to align
        H, W = dsize[1], dsize[0]
        rng = np.random.default_rng(12345)
        cases = []
        for j in range(batch):
            m0 = np.zeros((H, W), np.uint8)
            m0[H // 3:H // 2, W // 5:2 * W // 5] = 1
            m1 = np.roll(m0, W // 8, axis=1)
            c = dict(ori_img=rng.integers(0, 256, (H, W, 3), dtype=np.uint8), coarse_input=rng.integers(0, 256, (H, W, 3), dtype=np.uint8),
                     ori_mask=m0, target_mask=m1 * 255, guidance_text="" if variant == "2d" else "object", draw_mask=None)
            if variant == "2d":
                c.update(use_auto_draw=True, reduce_inp_artifacts=True, cons_area=np.maximum(m0, m1))
            else:
                c.update(draw_mask=np.maximum(m0, m1), reduce_inp_artifacts=True, cons_area=np.maximum(m0, m1))
            cases.append(c)
        was = model.unet.use_graph
        model.unet.use_graph = False
        kw = dict(end_step=params["end_step"], num_step=params["num_step"], start_step=params["start_step"], end_scale=params["end_scale"], verbose=False)
        if batch > 1:
            model.FreeFine_generation_batch(cases, params["guidance_scale"], params["eta"], seeds=0, **kw)
        else:
            c = dict(cases[0])
            model.FreeFine_generation(c.pop("ori_img"), c.pop("ori_mask"), c.pop("coarse_input"), c.pop("target_mask"), c.pop("guidance_text"),
                                      params["guidance_scale"], params["eta"], seed=0, **kw, **c)
        if batch > 1:               # the shapes of a trailing one-case batch too (other partial batches take the deterministic rule on every rank)
            c = dict(cases[0])
            model.FreeFine_generation(c.pop("ori_img"), c.pop("ori_mask"), c.pop("coarse_input"), c.pop("target_mask"), c.pop("guidance_text"),
                                      params["guidance_scale"], params["eta"], seed=0, **kw, **c)
        model.unet.use_graph = was
        torch.cuda.synchronize()
    return FD.sync_tune_table(src)


def run(model, dst_base, batch=4, params=None, rank=0, world=1, check_exist=True, verbose=True, dsize=(512, 512), variant="2d", depth_model=None):
    """edit every case of the variant's annotation file that this rank owns; rank 0 writes the variant's result JSON
    (2d: annotations_2d.json -> generated_results_freefine_2d.json; 3d_depth: annotations.json -> generated_results_freefine_depth.json).
    Returns the merged result list (on every rank)."""
    from . import dist as FD
    V = VARIANTS[variant]
    params = dict(V["params"], **(params or {}))
    seed = params.pop("seed")
    dst_gen = osp.join(dst_base, V["gen_subdir"])
    os.makedirs(dst_gen, exist_ok=True)
    data = load_json(osp.join(dst_base, V["annotations"]))
    if data is None:
        raise FileNotFoundError(osp.join(dst_base, V["annotations"]))
    cl = CaseList(data, dst_gen, check_exist)
    mine = [cl[i] for i in FD.shard_indices(len(cl), rank, world)]
    results = []
    pending = []
    if world > 1:
        warm_and_sync(model, batch, params, dsize, variant)

    def flush():
        if not pending:
            return
        cases = [p[0] for p in pending]
        inputs = [p[1] for p in pending]
        if len(inputs) == 1:
            kw = {k: v for k, v in inputs[0].items() if k not in ("ori_img", "ori_mask", "coarse_input", "target_mask", "guidance_text")}
            imgs = [model.FreeFine_generation(inputs[0]["ori_img"], inputs[0]["ori_mask"], inputs[0]["coarse_input"], inputs[0]["target_mask"],
                                              inputs[0]["guidance_text"], params["guidance_scale"], params["eta"], end_step=params["end_step"],
                                              num_step=params["num_step"], start_step=params["start_step"], seed=seed,
                                              end_scale=params["end_scale"], verbose=False, **kw)]
        else:
            imgs = model.FreeFine_generation_batch(inputs, params["guidance_scale"], params["eta"], end_step=params["end_step"],
                                                   num_step=params["num_step"], start_step=params["start_step"], seeds=seed,
                                                   end_scale=params["end_scale"], verbose=False)
        for c, img in zip(cases, imgs):
            path = save_img(img, dst_gen, c["da_n"], c["ins_id"], c["edit_ins"])
            results.append(dict(c, gen_img_path=path, key=f'{c["da_n"]}/{c["ins_id"]}/{c["edit_ins"]}'))
        pending.clear()

    # the prefetch thread is HOST-ONLY (file reads, resizes): a new Python thread starts on device 0 with its own current stream, and device
    # work issued from it would also race the consumer's graph captures; the 3d_rgb variant's depth network + warp run HERE, per case
    loader = {"2d": load_case, "3d_depth": load_case_3d_depth, "3d_rgb": read_case_3d_rgb}[variant]
    try:
        for case, inputs, err in _prefetch(mine, dst_base, 2 * batch, dsize, loader):
            if err is None and inputs.get("_raw_3d_rgb"):
                try:
                    inputs = finish_case_3d_rgb(inputs, depth_model)
                except Exception as e:  # noqa: BLE001 -- reported with the case attached, like a failed read
                    err = e
            if err is not None:
                if verbose:
                    print(f'[geobench] skipped {case["da_n"]}/{case["ins_id"]}/{case["edit_ins"]}: {err}')
                continue
            pending.append((case, inputs))
            if len(pending) == batch:
                flush()
        flush()
        merged = FD.gather_results(results) if world > 1 else results
    finally:
        if world > 1:
            FD.restore_tuning()      # the freeze of warm_and_sync ends with the sharded run, also when it ends in an exception
    if rank == 0:
        final = list(cl.existing_results) + merged
        new_data = {}
        for it in final:
            it = {k: v for k, v in it.items() if k != "key"}
            new_data.setdefault(it["da_n"], {"instances": {}})["instances"].setdefault(it["ins_id"], {})[it["edit_ins"]] = it
        save_json(new_data, osp.join(dst_base, V["results"]))
        if verbose:
            print(f"Total images processed: {len(final)}")
    return merged


# ---------------------------------------------------------------------------------------------------------------------
# stage 1 of GeoBench-2D: object removal / background generation (freefine_batch_infer_bggen_2d.py) -- produces the
# Geo-Bench-2D/inp_img_{blended,no_blend}/<da>/<ins>/inp_img.png files the edit stage pastes the moved object onto
# ---------------------------------------------------------------------------------------------------------------------
GEOBENCH_BGGEN = dict(guidance_text="empty scene", guidance_scale=7.5, eta=1.0, end_scale=0.5, end_step=35, num_step=50, start_step=1)


class InpaintCaseList:
    """CustomDatasetInpaint (:39-87): one case per (image, instance), inputs taken from the instance's first edit entry"""

    def __init__(self, data, dst_dir, check_exist=True):
        self.cases, self.existing_results = [], []
        for da_n, da in data.items():
            for ins_id, cur in da.get("instances", {}).items():
                if not cur:
                    continue
                item = dict(da_n=da_n, ins_id=ins_id, **cur[next(iter(cur))])
                path = osp.join(dst_dir, str(da_n), str(ins_id), "inp_img.png")
                if check_exist and osp.exists(path):
                    self.existing_results.append(dict(item, gen_img_path=path))
                else:
                    self.cases.append(item)

    def __len__(self):
        return len(self.cases)

    def __getitem__(self, i):
        return self.cases[i]


def blend_with_original(ori_img, hole_mask, generated):
    """the driver's optional paste-back (:186-190, "from brushnet"): cv2.GaussianBlur(mask, (21,21), 0) -> sigma 3.5 (OpenCV's
    0.3*((k-1)*0.5-1)+0.8), radius 10, reflect-101 border; mask_np = 1-(1-m)(1-blur/255); out = ori*(1-mask_np) + gen*mask_np.
    The mask is {0,1} uint8, so blur/255 is ~0.004 at most: the blend is the hard mask to within half a percent -- kept as is."""
    from scipy import ndimage
    m = hole_mask.astype(np.float64)
    blur = np.rint(ndimage.gaussian_filter(m, sigma=(3.5, 3.5) + (0,) * (m.ndim - 2), mode="mirror", truncate=10 / 3.5)) / 255
    mask_np = 1 - (1 - m) * (1 - blur)
    return (ori_img * (1 - mask_np) + generated * mask_np).astype(generated.dtype)


def run_bggen(model, dst_base, blending=True, params=None, rank=0, world=1, check_exist=True, verbose=True, dsize=(512, 512), seed=None,
              batch=4, bench="2D"):
    """remove the annotated object of every (image, instance) with FreeFine_background_generation (model must carry the bg-gen hook:
    register_attention_control_4bggen), `batch` cases per UNet batch.  seed=None draws a fresh seed per case like the reference (:162)."""
    import random
    from PIL import Image
    from src.utils.vis_utils import read_and_resize_mask_with_dilation
    from . import dist as FD
    params = dict(GEOBENCH_BGGEN, **(params or {}))
    # bench = "2D" | "3D": annotations_2d.json -> Geo-Bench-2D/..., annotations_3d.json -> Geo-Bench-3D/... (freefine_batch_infer_bggen_3d.py)
    ann = f"annotations_{bench.lower()}.json"
    out_dir = osp.join(dst_base, f"Geo-Bench-{bench}", "inp_img_blended" if blending else "inp_img_no_blend")
    os.makedirs(out_dir, exist_ok=True)
    data = load_json(osp.join(dst_base, ann))
    if data is None:
        raise FileNotFoundError(osp.join(dst_base, ann))
    cl = InpaintCaseList(data, out_dir, check_exist)
    mine = [cl[i] for i in FD.shard_indices(len(cl), rank, world)]
    done = []
    kw = dict(end_step=params["end_step"], num_step=params["num_step"], start_step=params["start_step"], end_scale=params["end_scale"],
              verbose=False)
    for b0 in range(0, len(mine), batch):
        group = mine[b0:b0 + batch]
        imgs = [read_and_resize_img(c["ori_img_path"], dsize) for c in group]
        holes = [read_and_resize_mask_with_dilation(c["ori_mask_path"], dsize, dilation_factor=30, forbit_area=None) for c in group]
        seeds = [(random.randint(0, 10 ** 16) if seed is None else seed) % (2 ** 63) for _ in group]
        if len(group) == 1:
            gens = [model.FreeFine_background_generation(imgs[0], holes[0], params["guidance_text"], params["guidance_scale"], params["eta"],
                                                         seed=seeds[0], **kw)]
        else:
            gens = model.FreeFine_background_generation_batch(
                [dict(ori_img=im, ori_mask=h, guidance_text=params["guidance_text"]) for im, h in zip(imgs, holes)],
                params["guidance_scale"], params["eta"], seeds=seeds, **kw)
        for c, im, h, gen in zip(group, imgs, holes, gens):
            if blending:
                gen = blend_with_original(im, h, gen)
            d = osp.join(out_dir, str(c["da_n"]), str(c["ins_id"]))
            os.makedirs(d, exist_ok=True)
            Image.fromarray(gen).save(osp.join(d, "inp_img.png"))
            done.append(dict(key=f'{c["da_n"]}/{c["ins_id"]}', da_n=c["da_n"], ins_id=c["ins_id"], inp_img_path=osp.join(d, "inp_img.png")))
    merged = FD.gather_results(done) if world > 1 else done
    if verbose and rank == 0:
        print(f"background images generated: {len(merged)} (+{len(cl.existing_results)} existing)")
    return merged


def make_synthetic_dataset(root, n_images=2, edits_per_image=2, size=96, seed=0, with_backgrounds=True, with_3d=False):
    """a GeoBenchMeta-shaped tree with seeded random images / rectangular instance masks / affine edit parameters (no dataset
    exists offline): annotations_2d.json, source PNGs, Geo-Bench-2D/inp_img_blended/<da>/<ins>/inp_img.png."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    data = {}
    for d in range(n_images):
        da = f"{d:04d}"
        img = rng.integers(0, 256, (size, size, 3), dtype=np.uint8)
        os.makedirs(osp.join(root, "source", da), exist_ok=True)
        ip = osp.join(root, "source", da, "img.png")
        Image.fromarray(img).save(ip)
        r0, c0 = int(rng.integers(size // 8, size // 3)), int(rng.integers(size // 8, size // 3))
        mask = np.zeros((size, size), np.uint8)
        mask[r0:r0 + size // 4, c0:c0 + size // 4] = 255
        mp = osp.join(root, "source", da, "mask_0.png")
        Image.fromarray(mask).save(mp)
        if with_backgrounds:
            inp_dir = osp.join(root, INP_SUBDIR, da, "0")
            os.makedirs(inp_dir, exist_ok=True)
            Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8)).save(osp.join(inp_dir, "inp_img.png"))
        edits = {}
        for e in range(edits_per_image):
            dx, dy = int(rng.integers(4, size // 4)), int(rng.integers(-4, size // 6))
            edits[str(e)] = dict(ori_img_path=ip, ori_mask_path=mp, edit_param=[dx, dy, 0, 0, 0, float(rng.integers(-20, 20)), 1.0, 1.0, 1.0],
                                 edit_prompt="move")
            if with_3d:      # the GeoBench-3D annotation fields + the pre-rendered coarse edit (here: the same object shifted)
                tmask = np.roll(mask, (dy, dx), axis=(0, 1))
                draw = np.clip(ndimage_max(tmask, 9), 0, 255)
                tp = osp.join(root, "source", da, f"tgt_{e}.png")
                dp = osp.join(root, "source", da, f"draw_{e}.png")
                Image.fromarray(tmask).save(tp)
                Image.fromarray(draw).save(dp)
                cdir = osp.join(root, "coarse3d_depth_anything", da, "0")
                os.makedirs(cdir, exist_ok=True)
                coarse = np.where(tmask[:, :, None] > 0, np.roll(img, (dy, dx), axis=(0, 1)), img)
                Image.fromarray(coarse).save(osp.join(cdir, f"{e}.png"))
                edits[str(e)].update(target_mask_0=tp, draw_mask=dp, obj_label="a cup")
        data[da] = {"instances": {"0": edits}}
    save_json(data, osp.join(root, "annotations_2d.json"))
    if with_3d:
        save_json(data, osp.join(root, "annotations.json"))
    return data


def ndimage_max(mask, k):
    from scipy import ndimage
    return ndimage.maximum_filter(mask, size=(k, k), mode="constant", cval=0)
