"""Drop-in for the reference's `src/demo/model.py` import path (freefine_batch_infer_2d.py:5): same names, MI355X engine."""
from freefine_amd.pipeline import FreeFine, FreeFinePipeline, seed_everything  # noqa: F401
from freefine_amd.scheduler import DDIMScheduler  # noqa: F401
