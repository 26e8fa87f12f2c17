"""CPU: the model-free arithmetic of the GeoBench metric suite (freefine_amd/metrics.py) against the numbers the REFERENCE's own functions produced
(tests/golden/g11_metrics.npz: evaluation/metrics/wrap_error.py, FID/fid_score.py:calculate_frechet_distance, FID/mmd.py, imported by tools/gen_golden.py run_g11)."""
import os

import numpy as np

from freefine_amd import metrics as FM

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_metrics.npz"))


def test_frechet_distance_and_feature_statistics():
    a, b = G["feat_a"], G["feat_b"]
    mu1, s1 = FM.feature_statistics(a)
    mu2, s2 = FM.feature_statistics(b)
    assert abs(FM.frechet_distance(mu1, s1, mu2, s2) - G["frechet"][0]) < 1e-9
    assert abs(FM.frechet_distance(mu1, s1, mu1, s1) - G["frechet"][1]) < 1e-9


def test_polynomial_mmd_and_kernel_distance():
    a, b = G["feat_a"], G["feat_b"]
    assert abs(FM.polynomial_mmd2(a[:200], b[:200]) - G["mmd2"][0]) < 1e-10
    assert abs(FM.polynomial_mmd2(a[:64], a[64:128]) - G["mmd2"][1]) < 1e-10
    np.random.seed(5)                                     # the reference draws its subsets from numpy's global generator (mmd.py:9)
    kd = FM.kernel_distance(a, b, n_subsets=7, subset_size=100)
    assert np.allclose(kd, G["kd"], atol=1e-10)


def test_warp_error_over_a_result_tree():
    data, k = {}, 0
    for d in range(2):
        inst = {}
        for e in range(2):
            inst[str(e)] = {nm: (k, nm) for nm in ("coarse_input_path", "gen", "tgt_mask_path")}
            k += 1
        data[str(d)] = {"instances": {"0": inst}}
    reader = lambda key: G[f"we_{key[0]}_{key[1]}"]
    assert abs(FM.calculate_we(data, "gen", reader=reader) - G["we"][0]) < 1e-12
    one = FM.warp_error(G["we_0_coarse_input_path"], G["we_0_gen"], G["we_0_tgt_mask_path"])
    assert 0 < one < 1 and FM.warp_error(G["we_0_gen"], G["we_0_gen"], G["we_0_tgt_mask_path"]) == 0
