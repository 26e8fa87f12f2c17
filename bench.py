"""bench.py -- the FreeFine hot path on MI355X, measured on BASELINE.json's metric and config.

    python bench.py --gpus N --steps K --warmup W        (N>1: one rank per GPU over RCCL -- either started by torch.distributed.run, or, when
                                                          no rank environment is set, bench.py starts the N ranks itself as child processes)

One "step" = one pass of the hot path over one batch of synthetic inputs = `--concurrent` x `--batch` full `FreeFine_generation`
edits at 512x512 on SD-2.1-base topology (865.9 M parameter UNet + SD VAE, seeded synthetic weights: no checkpoints exist
offline), N=50 DDIM schedule run in full per image (start_step=0: 50 inversion forwards with 2 UNet rows per image + 50
guided-denoising forwards with 4 logical UNet rows per image, TCA attention injection in blocks 10-15, local cross-attention,
masked CFG, masked DDPM step) + VAE encode/decode.  `--batch K` independent edits (own images, masks, seeds) share one
image-major UNet batch (FreeFine_generation_batch); `--concurrent C` such batches run on C HIP streams over the
same weights.  value = images / second.  Inputs are host uint8 images (768 KB each; the PCIe upload is inside the timed region,
as it is for the reference).
Rank r edits its own images (weak scaling: independent units, no data-path collective); value = images of
all ranks / max-over-ranks time.

The HEADLINE mode is `--dtype bf16x3` (split-bf16: fp32 storage, every GEMM / attention product on three bf16 MFMAs per term): the fastest
mode that holds the north star's 1e-3 latent tolerance on the metric's own schedules -- tests/test_pipeline_gpu.py gates it (and the f32 mode)
at 1e-3 ABSOLUTE against reference-generated N = 50 trajectories (G9: start_step 0 / 35 / 15 / 1) and against full-size SD-2.1 oracle
trajectories at N = 50 (G10: edit hook at start_step 35 and 0, bg-gen hook at 1, compose hook at 15).  The bf16 mode (4.4 images/s in round 3) does NOT hold that tolerance and is reported under
`fast_modes`, never as `value`.  Weights: seeded default-init tensors + freefine_amd.weights.plant_denoiser_path, so that the 50-step
trajectory is a denoising one with O(1) latents (with purely random weights nothing removes the DDPM noise and |latent| grows 14.6x).
The text encoder inside the timed region is a real-size CLIP-shaped transformers model on the device; its prompt cache is emptied before every
batch of edits, so every step pays for its prompts.

Extra objects on the JSON line:
  parity        this run's own check: the headline mode's latent trajectory against the f32 mode's over the FULL schedule (absolute L-inf,
                `passes`), beside the names of the gates that pin both modes to the reference / oracle.
  fast_modes    bf16 (and with --fp8-leg the e4m3-convolution variant): throughput by the same protocol at the headline's own layout (--batch images per UNet batch; the f32 leg at --extra-batch, like
                parity.f32_mode: each record names its batch x streams) + the deviation that disqualifies it.
  roofline      the dominant kernel (largest total time of an eagerly executed, HIP-event-timed image in this very process):
                achieved = algorithmic FLOPs per launch / average launch duration; peak = dense MFMA peak of the dtype.
  cpu_baseline  the CPU oracle (oracle/, "port") timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "bf16x3": 2500.0 / 3}   # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md (bf16x3: three bf16 MFMAs per product term)
PEAK_HBM_GBS = 8000.0
_F32_TRAJ = None          # the f32 mode's latent trajectory of the parity leg (reference of every deviation figure of the line)
_TEXT_ENCODER = None      # the real-size CLIP-shaped text encoder, built once per process
F_UNET = 0.804e12      # algorithmic FLOPs, one sample, one UNet forward @64x64
F_TCA = 72.5e9         # one extra attention pass in blocks 10-15 per sample-forward
F_VAE = 7.26e12        # 2 encodes + 2 decodes @512^2
F_VAE_ENC, F_VAE_DEC = 1.117e12, 2.515e12
F_UNET_PHASE_A = 0.4396e12   # conv_in + down blocks + mid block + up_blocks[0..1] of one sample-forward (FlopCounterMode on the oracle UNet)
F_UP2X = 37.75e9        # what the sub-pixel form of the three up-sampling convolutions does not execute per sample-forward (5/9 of 67.95 GFLOP);
F_UP2X_B = 16.78e9      # ... of a reference row in phase B (only up_blocks[2]'s upsampler); the VAE decoder's three: 386.5 GFLOP per decode
F_VAE_UP2X = 386.5e9
F_REF_TAIL = 57.8e9     # what a reference row runs behind the K / V projection of transformer block 15: both TCA passes of that block's self attention
                        # (2 x 21.5 GFLOP at S = 4096, C = 320) + to_out, cross attention, feed-forward, proj_out, conv_out (14.8 GFLOP)


def synth_inputs(idx=0):
    """SURVEY 8(d) synthetic inputs: seeded images, rectangular masks (non-wrapping draw-mask branch)."""
    ori_img = np.random.default_rng(2 * idx).integers(0, 256, (512, 512, 3), dtype=np.uint8)
    coarse = np.random.default_rng(2 * idx + 1).integers(0, 256, (512, 512, 3), dtype=np.uint8)
    ori_mask = np.zeros((512, 512), np.uint8)
    ori_mask[200:300, 100:200] = 1
    tgt_mask = np.zeros((512, 512), np.uint8)
    tgt_mask[200:300, 160:260] = 255
    draw = np.zeros((512, 512), np.uint8)
    draw[190:310, 150:270] = 1
    return ori_img, ori_mask, coarse, tgt_mask, draw


def add_sibling(model):
    """a second pipeline over the same weights with its own controller (for concurrent edits on another HIP stream)"""
    from freefine_amd.attention import Attention_Modulator, register_attention_control
    sib = model.share()
    sib.controller = Attention_Modulator(start_layer=10)
    register_attention_control(sib, sib.controller)
    sib.modify_unet_forward()
    return sib


def build_model(args, device, rank, world):
    from freefine_amd.attention import Attention_Modulator, register_attention_control
    from freefine_amd.config import UNetConfig, VAEConfig
    from freefine_amd.pipeline import FreeFinePipeline
    from freefine_amd.scheduler import DDIMScheduler
    from freefine_amd.text import ByteTokenizer, SyntheticTextEncoder, clip_shaped_text_encoder
    from freefine_amd.weights import plant_denoiser_path, synthetic_state, unet_param_shapes, vae_param_shapes
    ucfg, vcfg = UNetConfig.preset(args.model), VAEConfig.preset(args.vae)

    def unet_state():       # seeded default-init tensors + the planted denoiser path (O(1) latents over the 50-step schedule, see weights.py)
        st = synthetic_state(unet_param_shapes(ucfg), 0)
        planted = getattr(args, "planted", 0.0)          # (tools/* build models through this function with their own argument sets)
        return plant_denoiser_path(st, ucfg, planted) if planted > 0 else st
    if world > 1:
        from freefine_amd import dist as FD
        # rank 0 generates (stands for: reads) the weights, the others receive them over RCCL straight into device memory; bf16 payload
        # for the matrices in fast mode (the packers round them to bf16 anyway: packed weights are bit-identical on every rank)
        mdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        ust = FD.broadcast_state(unet_state() if rank == 0 else None, unet_param_shapes(ucfg), device, matrix_dtype=mdt)
        vst = FD.broadcast_state(synthetic_state(vae_param_shapes(vcfg), 1) if rank == 0 else None, vae_param_shapes(vcfg), device, matrix_dtype=mdt)
    else:
        ust, vst = unet_state(), synthetic_state(vae_param_shapes(vcfg), 1)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    global _TEXT_ENCODER
    if getattr(args, "text", "table") == "clip":
        if _TEXT_ENCODER is None:       # seeded: identical on every rank; shared by every mode built in this process
            _TEXT_ENCODER = clip_shaped_text_encoder(ucfg.cross_attention_dim).to(device)
        enc = _TEXT_ENCODER
    else:
        enc = SyntheticTextEncoder(ucfg.cross_attention_dim)
    model = FreeFinePipeline.from_state(ucfg, ust, vcfg, vst, ByteTokenizer(), enc, None, dtype, device,
                                        x3=args.dtype == "bf16x3", fp8_conv=bool(getattr(args, "fp8_conv", False)) and args.dtype == "bf16")
    model.text_cache = True             # ... but edit_once() empties the cache before every batch: each edit pays the encoder for its own prompts
    model.scheduler = DDIMScheduler.from_config(model.scheduler.config)
    controller = Attention_Modulator(start_layer=10)
    model.controller = controller
    register_attention_control(model, controller)
    model.modify_unet_forward()
    model.unet.use_graph = not args.no_graph
    model.dedup_rows = not args.no_dedup
    return model


def edit_once(model, args, idx):
    K = getattr(args, "batch", 1)
    model._text_cache.clear()           # no embedding survives from one batch of edits to the next: the text encoder runs inside every timed step
    if K > 1:       # image-level batching: K independent edits (own images / seeds) in one image-major UNet batch
        cases = []
        for j in range(K):
            ori_img, ori_mask, coarse, tgt_mask, draw = synth_inputs(idx * K + j)
            cases.append(dict(ori_img=ori_img, ori_mask=ori_mask, coarse_input=coarse, target_mask=tgt_mask,
                              guidance_text="a photo of a cup", draw_mask=draw))
        return model.FreeFine_generation_batch(cases, 7.5, 1.0, end_step=args.num_step, num_step=args.num_step, start_step=args.start_step,
                                               method_type="tca", seeds=[42 + j for j in range(K)], end_scale=0.0)[0]
    ori_img, ori_mask, coarse, tgt_mask, draw = synth_inputs(idx)
    return model.FreeFine_generation(ori_img, ori_mask, coarse, tgt_mask, "a photo of a cup", 7.5, 1.0, end_step=args.num_step,
                                     num_step=args.num_step, start_step=args.start_step, method_type="tca", verbose=True, seed=42,
                                     draw_mask=draw, end_scale=0.0)


def roofline_leg(model, args):
    """one image executed eagerly with a HIP event pair around every launch (ops.profile_*), on the launch stream."""
    from freefine_amd import ops
    was = model.unet.use_graph
    model.unet.use_graph = False
    ops.profile_begin()
    edit_once(model, args, 0)
    prof = ops.profile_end()
    model.unet.use_graph = was
    name, d = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
    avg_ms = d["total_ms"] / d["calls"]
    achieved = d["flops"] / d["calls"] / (avg_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[args.dtype]
    total_ms = sum(v["total_ms"] for v in prof.values())
    # every launch of the path is in the table (GEMM / attention with FLOPs; norms, layout and scheduler kernels with bytes only)
    table = sorted(((k, v["calls"], v["total_ms"], v["flops"] / max(v["total_ms"], 1e-9) / 1e9, v["bytes"] / max(v["total_ms"], 1e-9) / 1e6)
                    for k, v in prof.items()), key=lambda t: -t[2])
    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (tools/pmc.sh; FETCH_SIZE x2 per the gfx950
    # correction of MI355X_MICROARCH.md, WRITE_SIZE), recorded per round under profiles/: reported only if it is the same kernel
    traffic, traffic_note = None, None
    import glob
    for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_dominant.json")), reverse=True):     # newest round first
        recs = json.load(open(pmc))
        for rec in (recs if isinstance(recs, list) else [recs]):
            if traffic is None and rec.get("kernel") == name:
                traffic, traffic_note = rec["traffic_bytes_per_launch"], rec.get("note")
    out = dict(bound="mfma", kernel=name, achieved=round(achieved, 2), peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 4),
               traffic=traffic, launches=d["calls"], avg_launch_us=round(avg_ms * 1e3, 2),
               share_of_timed_kernels=round(d["total_ms"] / total_ms, 3))
    if traffic_note:
        out["traffic_note"] = traffic_note
    targs = name[name.find("<") + 1:name.rfind(">")].split(", ") if "igemm_pp_kernel<" in name else []
    if len(targs) > 5 and targs[5] == "true":     # SPLIT: one event-timed launch = two kernels
        out["launch_note"] = ("split-K launch: avg_launch_us brackets the slab kernel AND igemm_splitk_reduce_kernel (rocprofv3 lists them separately: "
                              "their two averages add up to this figure; achieved = algorithmic FLOPs / that sum)")
    # the next kernels by total time, same definitions (the first two trade places from run to run: 10.6 % vs 10.7 % of the timed kernels)
    out["next_kernels"] = [dict(kernel=k, achieved=round(tf, 2), frac=round(tf / peak, 4), share_of_timed_kernels=round(ms / total_ms, 3))
                           if tf > 0 else
                           dict(kernel=k, bound="hbm", achieved=round(gbs, 1), unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4),
                                share_of_timed_kernels=round(ms / total_ms, 3))
                           for k, c, ms, tf, gbs in table[1:6] if k != name]
    # the HBM-bound part of the path: algorithmic bytes (one read + one write of the tensor) per time, all such kernels together
    hb = [(v["total_ms"], v["bytes"]) for v in prof.values() if v["flops"] == 0 and v["bytes"] > 0]
    if hb:
        ms, by = sum(t for t, _ in hb), sum(b for _, b in hb)
        out["hbm_bound_kernels"] = dict(share_of_timed_kernels=round(ms / total_ms, 3), achieved=round(by / ms / 1e6, 1), unit="GB/s",
                                        peak=PEAK_HBM_GBS, frac=round(by / ms / 1e6 / PEAK_HBM_GBS, 4),
                                        note="GroupNorm, LayerNorm, concat, layout and scheduler kernels; algorithmic bytes = one read + one write")
    return out, table


def one_image_trajectory(model, args):
    """the latent trajectory of ONE edit of the workload (image 0, seed 42) over the full schedule: [n + 1, 2, 4, 64, 64] on the host"""
    ori_img, ori_mask, coarse, tgt_mask, draw = synth_inputs(0)
    model.FreeFine_generation(ori_img, ori_mask, coarse, tgt_mask, "a photo of a cup", 7.5, 1.0, end_step=args.num_step, num_step=args.num_step,
                              start_step=args.start_step, method_type="tca", verbose=False, seed=42, draw_mask=draw, end_scale=0.0,
                              return_intermediates=True)
    return torch.stack([t.float() for t in model.last_intermediates]).cpu()


def deviation(traj, ref):
    d = (traj - ref).abs().flatten(1).max(dim=1).values
    amax = ref.abs().max().item()
    return {"max_over_steps": float(f"{d.max().item():.3e}"), "final": float(f"{d[-1].item():.3e}"), "worst_step": int(d.argmax()),
            "relative_to_latent_abs_max": float(f"{d.max().item() / amax:.3e}")}


def timed_mode(args, device, mode, steps, fp8=False, base=7000, batch=None):
    """throughput of another arithmetic mode by the headline's protocol: --extra-batch images per UNet batch, same number of HIP streams, one warm-up
    pass per stream (tuning, graph capture), then `steps` timed steps between synchronisations.  Returns (record, model)."""
    import copy
    import threading
    a = copy.copy(args)
    a.dtype, a.fp8_conv = mode, fp8
    # side legs: the f32 leg at most --extra-batch images per UNet batch (the record says which), to bound the run time; the bf16 fast mode at the
    # headline's own layout (batch = args.batch), so that the two throughputs are like for like
    a.batch = batch if batch is not None else min(args.batch, args.extra_batch)
    m = build_model(a, device, 0, 1)
    models = [m] + [add_sibling(m) for _ in range(args.concurrent - 1)]
    streams = [torch.cuda.Stream(device=device) for _ in models]
    for j, (mm, st) in enumerate(zip(models, streams)):
        with torch.cuda.stream(st):
            edit_once(mm, a, base + 10 * j)
        st.synchronize()

    def worker(j):
        torch.cuda.set_device(device)
        with torch.cuda.stream(streams[j]):
            for i in range(steps):
                edit_once(models[j], a, base + 100 + 10 * j + i)
        streams[j].synchronize()
    torch.cuda.synchronize()
    t0 = time.time()
    th = [threading.Thread(target=worker, args=(j,)) for j in range(len(models))]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.time() - t0
    v = steps * len(models) * a.batch / dt
    n = args.num_step - args.start_step
    f_img = n * (2 * F_UNET + 4 * F_UNET + 4 * F_TCA) + F_VAE
    rec = {"value": round(v, 4), "unit": "images/s", "steps": steps, "images_per_unet_batch": a.batch, "concurrent_streams": len(models),
           "whole_path_frac_of_mfma_peak": round(f_img * v / 1e12 / PEAK_TFLOPS[mode], 4), "mfma_peak_tflops": round(PEAK_TFLOPS[mode], 1)}
    del models[1:]
    return rec, m


def parity_leg(args, device, model):
    """This run's own parity evidence for the HEADLINE mode: its latent trajectory over the FULL schedule of the workload (one image, same
    seed and noise) against the f32 mode's (exact-fp32 MFMA chain, itself <= 2e-5 from the oracle on the full-size fixtures), ABSOLUTE L-inf
    at every step, `passes` = max <= 1e-3; and the f32 mode's own throughput by the headline's protocol."""
    global _F32_TRAJ
    out = {"tolerance_latent_linf": 1e-3,
           "gates": ("tests/test_pipeline_gpu.py::test_metric_schedules_n50_vs_reference_golden (tiny topology, N=50, start_step 0/35/15, bg-gen 1, "
                     "compose 15: reference-generated, f32 + bf16x3), ::test_full_size_n50_schedules_vs_oracle_fixture (SD-2.1 topology 64x64, N=50, "
                     "start_step 35 and 0, N=20: oracle-generated, f32 + bf16x3) and ::test_full_size_n50_other_hooks_vs_oracle_fixture (the same topology "
                     "under the bg-gen hook at start_step 1 and the compose hook with R = 2 references at start_step 15), absolute latent L-inf <= 1e-3 at "
                     "every step; ::test_full_size_bench_layout_24_edits_per_batch_vs_oracle_fixture (this layout itself: 24 edits per UNet batch, split-bf16, "
                     "N=50 start_step 0, image 0 against the oracle fixture); profiles/r5_planted_gain_sensitivity.txt: the headline mode's deviation over this schedule at planted gains 0 / 0.5 / 1 / 3 "
                     "(2-3e-5 of |latent| max at every gain; bf16: 1-2.5e-2)")}
    rec, m32 = timed_mode(args, device, "f32", max(1, args.extra_steps // 3))      # (0.4-0.5 images/s: one timed step of the headline's batch layout)
    _F32_TRAJ = ref = one_image_trajectory(m32, args)
    del m32
    torch.cuda.empty_cache()
    out["latent_abs_max"] = round(ref.abs().max().item(), 3)
    out["latent_std_start_end"] = [round(ref[0, 0].std().item(), 3), round(ref[-1, 0].std().item(), 3)]
    out["schedule"] = f"N={args.num_step}, start_step={args.start_step}, one image, seed 42, planted denoiser gain {args.planted}"
    if args.dtype != "f32":
        dev = deviation(one_image_trajectory(model, args), ref)
        dev["passes"] = bool(dev["max_over_steps"] <= 1e-3)
        out[f"{args.dtype}_vs_f32_latent_linf"] = dev
        out["headline_passes_tolerance"] = dev["passes"]
    out["f32_mode"] = rec
    return out


def fast_modes_leg(args, device):
    """the modes that do NOT hold the tolerance, by the headline's protocol, each with the deviation that disqualifies it"""
    out = {}
    modes = [("bf16", False)] + ([("bf16+fp8_conv", True)] if args.fp8_leg else [])
    for name, fp8 in modes:
        rec, m = timed_mode(args, device, "bf16", args.extra_steps, fp8=fp8, base=8000, batch=args.batch)
        if _F32_TRAJ is not None:
            dev = deviation(one_image_trajectory(m, args), _F32_TRAJ)
            dev["passes"] = bool(dev["max_over_steps"] <= 1e-3)
            rec["latent_linf_vs_f32"] = dev
        rec["what"] = ("bf16 storage and MFMA operands, fp32 accumulate / softmax / statistics" +
                       ("; the two 3x3 convolutions of every ResBlock on e4m3 operands (v_mfma_f32_16x16x32_fp8_fp8)" if fp8 else ""))
        out[name] = rec
        del m
        torch.cuda.empty_cache()
    return out


def cpu_baseline_leg(args):
    """the oracle (CPU restatement, 'port') on this host: one inversion forward (B=2) + one guided forward (B=4, TCA +
    local cross-attention) of the same SD-2.1 topology at 64x64, then extrapolated to the schedule; VAE bracket timed once."""
    from oracle import attention_modulation as OA
    from oracle import sd_unet, sd_vae
    torch.set_num_threads(min(os.cpu_count(), args.cpu_threads))    # torch's CPU convs stop scaling (and regress) far below 256 threads
    cores = torch.get_num_threads()
    with torch.no_grad():
        net = sd_unet.init_unet(sd_unet.unet_config(args.model), seed=0, perturb_norms=False)
        D = net.cfg.cross_attention_dim
        g = torch.Generator().manual_seed(0)
        x2, e2 = torch.randn(2, 4, 64, 64, generator=g), torch.randn(2, 77, D, generator=g)
        t0 = time.time()
        net(x2, torch.tensor(481), e2)
        t_inv = time.time() - t0
        mod = OA.Modulator("edit", 32)
        net.set_modulator(mod)
        _, om, _, tm, _ = synth_inputs(0)
        mod.use_tca, mod.method, mod.layer_idx, mod.local_edit, mod.context_guidance = True, "tca", list(range(10, 16)), True, 0.5
        mod.fg_ref_mask, mod.fg_retain_mask, mod.local_edit_region = torch.tensor(om), torch.tensor(tm), torch.tensor(tm)
        x4, e4 = torch.randn(4, 4, 64, 64, generator=g), torch.randn(4, 77, D, generator=g)
        t0 = time.time()
        net(x4, torch.tensor(481), e4)
        t_den = time.time() - t0
        t_vae = 0.0
        if args.vae == "sd" and not args.cpu_skip_vae:
            vae = sd_vae.init_vae(sd_vae.vae_config("sd"), seed=1, perturb_norms=False)
            img = torch.randn(1, 3, 512, 512, generator=g)
            t0 = time.time()
            z = vae.encode_mean(img)
            vae.decode(z)
            t_vae = 2 * (time.time() - t0)     # two images are encoded and decoded per edit
    n = args.num_step - args.start_step
    per_image = n * (t_inv + t_den) + t_vae
    return dict(value=round(1.0 / per_image, 6), unit="images/s", cores=cores, host_cpu_count=os.cpu_count(), kind="port",
                sample=f"1 inversion UNet forward B=2 ({t_inv:.2f}s) + 1 guided forward B=4 with TCA ({t_den:.2f}s) at 64x64, x{n} steps; "
                       f"VAE encode+decode @512^2 once x2 ({t_vae:.2f}s); torch fp32, {cores} threads")


def launch_command(n, argv, port=None):
    """`python bench.py --gpus N` without a rank environment: the command line of the N-rank job (one process per GPU over RCCL,
    rendezvous on 127.0.0.1) -- what the reference's run_script_2D.sh:12-14 does with `torchrun --nproc_per_node=8`."""
    port = port or int(os.environ.get("MASTER_PORT", 0)) or (29500 + os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def maybe_launch_ranks(args, argv, environ=os.environ):
    """--gpus N > 1 outside a torch.distributed.run job: start the N rank processes as CHILDREN (this parent has not touched the GPU and
    never will), relay their output -- rank 0 prints the JSON line -- and return the job's exit code; None when this process is itself a
    rank (WORLD_SIZE set: the driver's own `torch.distributed.run ... bench.py --gpus N` form, and our children) or N == 1."""
    if args.gpus <= 1 or "WORLD_SIZE" in environ:
        return None
    import subprocess
    return subprocess.run(launch_command(args.gpus, argv), env=dict(environ)).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU); > 1 without a rank environment starts them through torch.distributed.run")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tune-file", default=os.environ.get("FFN_IGEMM_TUNE_FILE", ""),
                    help="igemm tuning table: loaded before the warm-up if it exists, written after it (profiling runs then skip the tuner's candidate launches)")
    ap.add_argument("--dtype", default="bf16x3", choices=["bf16", "f32", "bf16x3"],
                    help="arithmetic mode of the timed region; default = the fastest mode that holds the 1e-3 latent tolerance on the N=50 fixtures")
    ap.add_argument("--planted", type=float, default=3.0, help="gain of the planted denoiser path of the synthetic UNet weights (0 = purely random weights)")
    ap.add_argument("--text", default="clip", choices=["clip", "table"], help="text encoder inside the timed region: real-size CLIP-shaped transformers model "
                    "on the device (one prompt per call, cache emptied before every batch of edits) or the table-lookup stand-in of the tests")
    ap.add_argument("--extra-steps", dest="extra_steps", type=int, default=3, help="timed steps of the f32 / fast-mode legs")
    ap.add_argument("--model", default="sd21-base")
    ap.add_argument("--vae", default="sd")
    ap.add_argument("--num-step", dest="num_step", type=int, default=50)
    ap.add_argument("--start-step", dest="start_step", type=int, default=0)
    ap.add_argument("--concurrent", type=int, default=2)
    ap.add_argument("--extra-batch", dest="extra_batch", type=int, default=8, help="images per UNet batch of the f32 / fast-mode side legs (min with --batch)")
    ap.add_argument("--batch", type=int, default=24, help="independent edits per UNet batch (image-level batching).  24 x 2 streams: the 32x32 level's inversion "
                    "batches fill 256 CUs without split-K (profiles/r5_batch_x_streams.txt: 8 x 2 1.91, 16 x 2 1.97, 16 x 3 2.01, 24 x 2 2.02 images/s); a step is 48 images, "
                    "the driver's --steps 20 --warmup 5 about 12 minutes with the side legs at --extra-batch")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-dedup", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-ref-layout", dest="no_ref_layout", action="store_true", help="skip the extra one-image-per-UNet-batch measurement")
    ap.add_argument("--ref-streams", dest="ref_streams", type=int, default=6, help="HIP streams of the one-image-per-UNet-batch measurement")
    ap.add_argument("--cpu-skip-vae", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--fp8-conv", dest="fp8_conv", action="store_true", help="bf16 mode with e4m3 ResBlock convolutions (FFN_FP8) as the timed configuration")
    ap.add_argument("--fp8-leg", dest="fp8_leg", action="store_true", help="also measure the bf16 + e4m3-convolution variant under fast_modes (it bought "
                    "nothing in the round-3 driver run, so it is off the default line)")
    ap.add_argument("--no-fp8-leg", dest="no_fp8_leg", action="store_true", help="accepted and ignored (round-3 profiling scripts): the fp8 leg is opt-in now")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity leg (f32 trajectory + throughput, the headline mode's latent deviation)")
    ap.add_argument("--no-fast-modes", dest="no_fast_modes", action="store_true", help="skip the bf16 fast-mode leg")
    args = ap.parse_args()
    rc = maybe_launch_ranks(args, sys.argv[1:])
    if rc is not None:
        sys.exit(rc)

    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    # smoke-testing the multi-rank path on a 1-GPU box: FFN_BENCH_SHARE_DEVICE=1 puts every rank on cuda:0 and uses gloo (RCCL refuses
    # two ranks on one device); the measured number is then meaningless, only the code path is exercised
    share = os.environ.get("FFN_BENCH_SHARE_DEVICE") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    from freefine_amd import _lib
    _lib.load()   # no fallback: fail loudly if the HIP extension is missing
    from freefine_amd import ops as _ops
    if args.tune_file:
        _ops.tune_table_load(args.tune_file)
    model = build_model(args, device, rank, world)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # `--concurrent C`: C independent edits in flight per GPU, each on its own HIP stream with its own pipeline state over the
    # SAME weights (GeoBench cases are independent units); one step = C images.  Graph capture happens in the sequential warm-up.
    import threading
    models = [model] + [add_sibling(model) for _ in range(args.concurrent - 1)]
    streams = [torch.cuda.Stream(device=device) for _ in models]
    outs = [None] * len(models)
    if world > 1:
        # every rank launches the SAME igemm configurations: one eager edit tunes every shape of the schedule, rank 0's table is
        # broadcast, and only then are the forward graphs captured (the tuner picks by timing, which may differ between GPUs;
        # identical tables make the bf16 results of the sharded run independent of which rank edited a case)
        from freefine_amd import dist as _fdist
        was = model.unet.use_graph
        model.unet.use_graph = False
        edit_once(model, args, rank * 1000 + 999)
        model.unet.use_graph = was
        torch.cuda.synchronize()
        _fdist.sync_tune_table(0)
    for j, (m, st) in enumerate(zip(models, streams)):
        with torch.cuda.stream(st):
            for i in range(max(args.warmup, 1 if args.concurrent > 1 else 0)):
                edit_once(m, args, rank * 1000 + 10 * j + i)
        st.synchronize()

    if args.tune_file and rank == 0:
        _ops.tune_table_save(args.tune_file)

    def worker(j):
        torch.cuda.set_device(device)
        with torch.cuda.stream(streams[j]):
            for i in range(args.steps):
                outs[j] = edit_once(models[j], args, rank * 1000 + 100 + 10 * j + i)
        streams[j].synchronize()

    barrier()
    t0 = time.time()
    if len(models) == 1:
        worker(0)
    else:
        th = [threading.Thread(target=worker, args=(j,)) for j in range(len(models))]
        [t.start() for t in th]
        [t.join() for t in th]
    barrier()
    dt = time.time() - t0
    out = outs[0]
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    assert out.shape == (512, 512, 3) and out.dtype == np.uint8

    if rank == 0:
        n = args.num_step - args.start_step
        f_img = n * (2 * F_UNET + 4 * F_UNET + 4 * F_TCA) + F_VAE
        # executed FLOPs: the exact (math-preserving) reductions that are ON lower what runs, not what the algorithm needs -- row
        # de-duplication (3 of the 4 guided rows are physical), decode of the edited latent only in the batched path
        rows_g = 3 if model.dedup_rows else 4
        # reference-stream reuse: each reference row skips conv_in ... up_blocks[1] (439.6 of the 804.3 GFLOP of a sample-forward)
        reuse_on = bool(getattr(model, "reuse_ref_stream", False)) and model.unet.reuse_replays > 0
        skipped = (rows_g - 2) * F_UNET_PHASE_A if reuse_on else 0.0
        if reuse_on and getattr(model, "drop_ref_tail", False):      # every guided step but the last: the reference rows stop after block 15's K / V
            skipped += (rows_g - 2) * F_REF_TAIL * (n - 1) / n
        n_dec = 1 if args.batch > 1 else 2
        f_exec = n * ((2 + rows_g) * F_UNET - skipped + rows_g * F_TCA) + 2 * F_VAE_ENC + n_dec * F_VAE_DEC
        # the state the models were actually built in (not the environment switch): sub-pixel form of nearest-2x + 3x3 conv
        up2x_unet = any(b.up2 is not None for b in model.unet.up)
        up2x_vae = any(b.up2 is not None for b in model.vae.dec_up)
        up2x_on = up2x_unet or up2x_vae
        if up2x_unet:           # nearest-2x + 3x3 conv evaluated as four 2x2 convolutions at low resolution: 4/9 of those FLOPs
            full_rows = 2 + (2 if reuse_on else rows_g)
            f_exec -= n * (full_rows * F_UP2X + ((rows_g - 2) * F_UP2X_B if reuse_on else 0.0))
        if up2x_vae:
            f_exec -= n_dec * F_VAE_UP2X
        value = world * args.steps * args.concurrent * args.batch / dt
        line = {
            "metric": "edited images/sec/GPU @512px 50-step DDIM", "value": round(value, 4), "unit": "images/s", "n_gpus": world,
            "value_scope": "whole job: images/s summed over the n_gpus ranks (max-over-ranks time); per GPU = value / n_gpus", "value_per_gpu": round(value / world, 4),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"SD-2.1-base topology ({args.model}) 512x512 FreeFine_generation edit, {args.num_step}-step DDIM schedule "
                                   f"(start_step={args.start_step}: per image {n} inversion forwards x 2 rows + {n} guided forwards x 4 rows, TCA blocks "
                                   f"10-15, masked CFG 7.5, eta=1) + VAE bracket; {args.batch} independent edits per UNet batch x "
                                   f"{args.concurrent} HIP streams; seeded random weights" + (f" + planted denoiser path (gain {args.planted})" if args.planted > 0 else ""),
                       "text_encoder": ("real-size CLIP-shaped transformers CLIPTextModel (23 layers, width 1024) on the device, one prompt per call, cache emptied before every batch of edits: "
                                        f"{model.text_encoder_calls} encoder calls by this pipeline so far") if args.text == "clip" else "table lookup stand-in",
                       "fp8_convolutions": bool(args.fp8_conv and args.dtype == "bf16"),
                       "images_per_gpu_per_step": args.concurrent * args.batch, "concurrent_streams": args.concurrent,
                       "images_per_unet_batch": args.batch, "unet_batch": 4 * args.batch, "hip_graph": not args.no_graph,
                       "exact_row_dedup": model.dedup_rows and "on: the duplicated reference row of the CFG batch is evaluated once (3 physical rows), outputs unchanged",
                       "reference_stream_reuse": ("on: the guided loop's reference row re-enters at up_blocks[2] from the state the inversion pass recorded for the "
                                                  "same (latent, timestep, prompt) and, at every step but the last, stops behind block 15's K / V projection (its eps is dead there); "
                                                  "outputs unchanged") if reuse_on else "off",
                       "subpixel_upsample_convs": (f"nearest-2x + 3x3 conv = four 2x2 convolutions at low resolution (4/9 of the FLOPs): UNet {'on' if up2x_unet else 'off'}, VAE decoder {'on' if up2x_vae else 'off'}") if up2x_on else "off",
                       "vae_decode": "batched path decodes the edited latent only (the reference decodes the reference stream too and drops it unless return_ori)" if args.batch > 1 else "both streams, like the reference",
                       "algorithmic_tflop_per_image": round(f_img / 1e12, 1),
                       "whole_path_tflops_per_gpu": round(f_img * value / world / 1e12, 1),
                       "whole_path_frac_of_mfma_peak": round(f_img * value / world / 1e12 / PEAK_TFLOPS[args.dtype], 4),
                       "executed_tflop_per_image": round(f_exec / 1e12, 1),
                       "executed_tflops_per_gpu": round(f_exec * value / world / 1e12, 1),
                       "executed_frac_of_mfma_peak": round(f_exec * value / world / 1e12 / PEAK_TFLOPS[args.dtype], 4)},
        }
        if args.batch > 1 and not args.no_ref_layout and world == 1:
            # the reference's own batch layout (one image per UNet call: inversion B=2, guided denoising B=4 -- BASELINE.json configs[1]
            # "batch=4"), same streams, measured beside the batched figure so both are on record
            import copy
            a1 = copy.copy(args)
            a1.batch = 1
            # one image per UNet call leaves most of the chip idle (48 tiles of the 64x64-level GEMMs for 256 CUs), so this layout runs more
            # HIP streams than the batched one (measured: 2 streams 2.04, 4: 2.01, 6: 2.37 images/s; the host threads are the limit)
            n1 = max(args.ref_streams, len(models))
            models1 = models + [add_sibling(model) for _ in range(n1 - len(models))]
            streams1 = streams + [torch.cuda.Stream(device=device) for _ in range(n1 - len(streams))]

            def worker1(j):
                torch.cuda.set_device(device)
                with torch.cuda.stream(streams1[j]):
                    for i in range(ref_steps):
                        edit_once(models1[j], a1, 5000 + 10 * j + i)
                streams1[j].synchronize()

            ref_steps = 1
            for j in range(n1):                 # warm-up: graphs / tuning of the 2- and 3-row shapes
                worker1(j)
            ref_steps = 3
            torch.cuda.synchronize()
            t1 = time.time()
            th = [threading.Thread(target=worker1, args=(j,)) for j in range(n1)]
            [t.start() for t in th]
            [t.join() for t in th]
            torch.cuda.synchronize()
            v1 = ref_steps * n1 / (time.time() - t1)
            line["config"]["reference_batch_layout"] = {
                "value": round(v1, 4), "unit": "images/s", "images_per_unet_batch": 1, "unet_batch": 4, "concurrent_streams": n1,
                "note": "same path, one image per UNet call (inversion B=2, guided B=4 logical / 3 physical rows; the reference row re-enters at up_blocks[2])"}
            del models1[len(models):], streams1[len(streams):]
            torch.cuda.empty_cache()
        if not args.no_roofline:
            line["roofline"], table = roofline_leg(model, args)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_kernel_table.txt"), "w") as f:
                f.write("kernel\tcalls\ttotal_ms\talgorithmic_TFLOP/s\talgorithmic_GB/s\n")
                for k, c, ms, gf, gbs in table:
                    f.write(f"{k}\t{c}\t{ms:.3f}\t{gf:.1f}\t{gbs:.1f}\n")
        if not args.no_parity and world == 1:
            line["parity"] = parity_leg(args, device, model)
        if not args.no_fast_modes and world == 1 and args.dtype != "bf16":
            line["fast_modes"] = fast_modes_leg(args, device)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline_leg(args)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
