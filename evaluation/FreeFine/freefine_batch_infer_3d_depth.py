"""GeoBench-3D (depth-guided coarse edits rendered beforehand) on the MI355X engine -- same entry point as the reference's
evaluation/FreeFine/freefine_batch_infer_3d_depth.py: reads <base-dir>/annotations.json and coarse3d_depth_anything/<da>/<ins>/<edit>.png,
writes Geo-Bench-3D/Gen_results_FreeFine_depth/... and generated_results_freefine_depth.json.

    python evaluation/FreeFine/freefine_batch_infer_3d_depth.py --base-dir <GeoBenchMeta> [--model ...] [--batch 4]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv += ["--variant", "3d_depth"]
import freefine_batch_infer_2d as drv  # noqa: E402

if __name__ == "__main__":
    drv.main()
