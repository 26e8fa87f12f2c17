"""Drop-in for the reference's `src/utils/attention.py` import path (freefine_batch_infer_2d.py:8)."""
from freefine_amd.attention import (Attention_Modulator, AttentionControl, AttentionStore, register_attention_control,  # noqa: F401
                                    register_attention_control_4bggen, register_attention_control_compose)
