"""The model-free arithmetic of the GeoBench metric suite (/root/reference/evaluation/metrics/): warp error, the Frechet distance between two
Gaussians of features, and the polynomial-kernel MMD^2 (the "kernel distance" of fid_kd.py).  The feature extractors the reference feeds them with
(Inception-v3, DINOv2, CLIP, HPSv2, ImageReward, DIFT) need model weights that do not exist offline and are NOT built; these functions take
features / images and are pinned to the reference's own functions by tests/golden/g11_metrics.npz (tools/gen_golden.py run_g11).
Host-side numpy: a metric pass is a few reductions over at most thousands of feature rows."""
import numpy as np


def warp_error(coarse, generated, mask):
    """one sample of wrap_error.py:calculate_we (:14-17): images uint8 / float HWC in [0, 255], mask HW in [0, 255];
    sum |coarse * m - generated * m| / sum(m) with m the mask / 255 repeated over the 3 channels"""
    a, b = np.asarray(coarse, np.float64) / 255, np.asarray(generated, np.float64) / 255
    m = np.repeat((np.asarray(mask, np.float64) / 255)[..., None], 3, axis=2)
    return float(np.abs(a * m - b * m).sum() / m.sum())


def calculate_we(data, image_label, reader=None):
    """wrap_error.py:calculate_we over a GeoBench result tree data[image]["instances"][instance][sample] with the paths
    coarse_input_path / <image_label> / tgt_mask_path; `reader` maps a path to an array (default: PIL)"""
    if reader is None:
        from PIL import Image
        reader = lambda p: np.array(Image.open(p))
    total, num = 0.0, 0
    for image in data.values():
        for instance in image["instances"].values():
            for sample in instance.values():
                total += warp_error(reader(sample["coarse_input_path"]), reader(sample[image_label]), reader(sample["tgt_mask_path"]))
                num += 1
    return total / num


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """FID/fid_score.py:calculate_frechet_distance (:146-199): ||mu1 - mu2||^2 + Tr(S1 + S2 - 2 sqrt(S1 S2)), with the eps-regularised retry when
    the matrix square root is not finite and the imaginary round-off dropped"""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        off = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + off).dot(sigma2 + off))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError(f"Imaginary component {np.max(np.abs(covmean.imag))}")
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


def feature_statistics(features):
    """mean and covariance of feature rows, as calculate_activation_statistics (:201-223) computes them (np.cov, rowvar=False)"""
    f = np.asarray(features, np.float64)
    return f.mean(axis=0), np.cov(f, rowvar=False)


def polynomial_mmd2(X, Y, degree=3, gamma=None, coef0=1.0):
    """FID/mmd.py:compute_polynomial_mmd (:24-35, 38-60): unbiased MMD^2 with k(x, y) = (gamma <x, y> + coef0)^degree, gamma = 1 / dim by default;
    X and Y hold the same number of rows"""
    X, Y = np.asarray(X, np.float64), np.asarray(Y, np.float64)
    m = X.shape[0]
    assert Y.shape[0] == m
    g = 1.0 / X.shape[1] if gamma is None else gamma
    k = lambda A, B: (g * (A @ B.T) + coef0) ** degree
    kxx, kyy, kxy = k(X, X), k(Y, Y), k(X, Y)
    sxx = kxx.sum() - np.trace(kxx)
    syy = kyy.sum() - np.trace(kyy)
    return float((sxx + syy) / (m * (m - 1)) - 2 * kxy.sum() / (m * m))


def kernel_distance(feat_real, feat_gen, n_subsets=100, subset_size=1000, rng=None):
    """FID/mmd.py:compute_mmd (:5-21): MMD^2 over random equally sized subsets (numpy's global generator unless `rng` is given); the reference
    reports the mean of the returned vector (fid_kd.py:39)"""
    rng = np.random if rng is None else rng
    m = min(min(feat_real.shape[0], feat_gen.shape[0]), subset_size)
    out = np.zeros(n_subsets)
    for i in range(n_subsets):
        g = feat_real[rng.choice(len(feat_real), m, replace=False)]
        r = feat_gen[rng.choice(len(feat_gen), m, replace=False)]
        out[i] = polynomial_mmd2(g, r)
    return out
