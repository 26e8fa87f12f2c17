"""GeoBench-2D stage 1 (object removal / background generation) on the MI355X engine -- same entry point as the reference's
evaluation/FreeFine/freefine_batch_infer_bggen_2d.py; writes <base-dir>/Geo-Bench-2D/inp_img_{blended,no_blend}/<da>/<ins>/inp_img.png.

    python evaluation/FreeFine/freefine_batch_infer_bggen_2d.py --base-dir <GeoBenchMeta> [--model ...] [--no-blending]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

from src.demo.model import DDIMScheduler, FreeFinePipeline  # noqa: E402
from src.utils.attention import Attention_Modulator, register_attention_control_4bggen  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base-dir", required=True)
    ap.add_argument("--model", default="synthetic:sd21-base")
    ap.add_argument("--dtype", default="bf16x3", choices=["bf16x3", "f32", "bf16"], help="bf16x3 = split-bf16 (default, bench.py's headline mode), f32 = parity mode, bf16 = fast mode")
    ap.add_argument("--no-blending", action="store_true")
    ap.add_argument("--bench", default="2D", choices=["2D", "3D"])
    ap.add_argument("--batch", type=int, default=4, help="cases processed together in one UNet batch")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank, local = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=device)
    model = FreeFinePipeline.from_pretrained(args.model, torch_dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float32, device=device, broadcast="auto", x3=args.dtype == "bf16x3").to(device)
    model._progress_bar_config = {"disable": True}
    model.scheduler = DDIMScheduler.from_config(model.scheduler.config)
    controller = Attention_Modulator(start_layer=10)
    model.controller = controller
    register_attention_control_4bggen(model, controller)
    model.modify_unet_forward()
    model.unet.use_graph = True
    from freefine_amd import geobench
    geobench.run_bggen(model, args.base_dir, blending=not args.no_blending, rank=rank, world=world, batch=args.batch, bench=args.bench)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
