"""GeoBench-3D object removal / background generation -- same entry point as the reference's
evaluation/FreeFine/freefine_batch_infer_bggen_3d.py (annotations_3d.json -> Geo-Bench-3D/inp_img_{blended,no_blend}/...)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv += ["--bench", "3D"]
import freefine_batch_infer_bggen_2d as drv  # noqa: E402

if __name__ == "__main__":
    drv.main()
