"""TEST INFRASTRUCTURE (oracle): fp64 restatement of the FSIMc index the reference's eval loop prints.

    test.py:95-96     fsim_index = piq.fsim(hr, sr, data_range=1., reduction='none'); avg_fsim += fsim_index.item()

`piq` is a third-party dependency that is absent from /root/reference and from this image (environment.yml:118 pins
piq==0.8.0).  This file restates the published algorithm -- Zhang, Zhang, Mou, Zhang, "FSIM: A Feature Similarity Index for
Image Quality Assessment", IEEE TIP 2011, with the phase congruency of Kovesi's phasecong2 as piq 0.8.0 implements it
(`piq/fsim.py`: fsim, _construct_filters, _phase_congruency; `piq/functional`: rgb2yiq, scharr_filter, gradient_map,
similarity_map, get_meshgrid, ifftshift) -- in numpy float64 with numpy's FFT.  PARITY UNPINNED: there is nothing here to
run piq against; the anchors in tests/test_metrics.py are identity -> 1, symmetry, invariance of the grey-level index to the
chroma terms on grey images, monotone degradation under noise / blur, and this restatement against an independent evaluation of
the filter bank through explicit DFT matrices.

Only tests/ may import this module.  The product path is m2trans_amd/metrics.py -> m2t_eval_fsim (csrc/k_fsim.hip)."""
from __future__ import annotations

import math

import numpy as np

EPS32 = float(np.finfo(np.float32).eps)      # piq takes torch.finfo(x.dtype).eps of its float32 inputs


def get_meshgrid(h: int, w: int):
    """piq.functional.get_meshgrid: normalised frequency coordinates, x along rows (dim 0), y along columns (dim 1)."""
    def axis(n):
        if n % 2:
            return np.arange(-(n - 1) / 2, n / 2) / (n - 1)
        return np.arange(-n / 2, n / 2) / n
    return np.meshgrid(axis(h), axis(w), indexing="ij")


def ifftshift(x: np.ndarray) -> np.ndarray:
    """piq.functional.ifftshift: roll by -(n // 2) along every axis."""
    return np.roll(x, [-(n // 2) for n in x.shape], axis=tuple(range(x.ndim)))


def lowpassfilter(h: int, w: int, cutoff: float = 0.45, n: int = 15) -> np.ndarray:
    gx, gy = get_meshgrid(h, w)
    radius = np.sqrt(gx ** 2 + gy ** 2)
    return ifftshift(1.0 / (1.0 + (radius / cutoff) ** (2 * n)))


def construct_filters(h: int, w: int, scales=4, orientations=4, min_length=6, mult=2, sigma_f=0.55, delta_theta=1.2) -> np.ndarray:
    """[orientations * scales][h][w] log-Gabor bank, index o * scales + s (piq _construct_filters)."""
    theta_sigma = math.pi / (orientations * delta_theta)
    gx, gy = get_meshgrid(h, w)
    radius = ifftshift(np.sqrt(gx ** 2 + gy ** 2))
    theta = ifftshift(np.arctan2(-gy, gx))
    radius[0, 0] = 1.0
    sintheta, costheta = np.sin(theta), np.cos(theta)
    lp = lowpassfilter(h, w)
    log_gabor = []
    for s in range(scales):
        omega0 = 1.0 / (min_length * mult ** s)
        f = np.exp(-(np.log(radius / omega0) ** 2) / (2 * math.log(sigma_f) ** 2)) * lp
        f[0, 0] = 0.0
        log_gabor.append(f)
    out = []
    for o in range(orientations):
        angl = o * math.pi / orientations
        ds = sintheta * math.cos(angl) - costheta * math.sin(angl)
        dc = costheta * math.cos(angl) + sintheta * math.sin(angl)
        spread = np.exp(-(np.abs(np.arctan2(ds, dc)) ** 2) / (2 * theta_sigma ** 2))
        for s in range(scales):
            out.append(spread * log_gabor[s])
    return np.stack(out)


def phase_congruency(lum: np.ndarray, scales=4, orientations=4, k=2.0, fft2=np.fft.fft2, ifft2=np.fft.ifft2) -> np.ndarray:
    """lum [h][w] (0..255) -> PC map [h][w] (piq _phase_congruency)."""
    h, w = lum.shape
    filt = construct_filters(h, w, scales, orientations)
    imfft = fft2(lum)
    filt_ifft = np.stack([ifft2(f).real for f in filt]) * math.sqrt(h * w)
    eo = np.stack([ifft2(imfft * f) for f in filt]).reshape(orientations, scales, h, w)
    even, odd = eo.real, eo.imag
    an = np.sqrt(even ** 2 + odd ** 2)
    fr = filt.reshape(orientations, scales, h, w)
    em_n = (fr[:, 0] ** 2).sum(axis=(-2, -1))                                    # [o]
    sum_e, sum_o = even.sum(axis=1), odd.sum(axis=1)                              # [o][h][w]
    x_energy = np.sqrt(sum_e ** 2 + sum_o ** 2) + EPS32
    mean_e, mean_o = sum_e / x_energy, sum_o / x_energy
    energy = (even * mean_e[:, None] + odd * mean_o[:, None] - np.abs(even * mean_o[:, None] - odd * mean_e[:, None])).sum(axis=1)
    e2 = (an[:, 0] ** 2).reshape(orientations, h * w)
    median_e2n = np.sort(e2, axis=1)[:, (h * w - 1) // 2]                         # torch.median: the lower median
    noise_power = (-median_e2n / math.log(0.5)) / em_n
    fi = filt_ifft.reshape(orientations, scales, h, w)
    sum_an2 = (fi ** 2).sum(axis=(1, 2, 3))
    sum_ai_aj = np.zeros(orientations)
    for s in range(scales - 1):
        sum_ai_aj += (fi[:, s:s + 1] * fi[:, s + 1:]).sum(axis=(1, 2, 3))
    noise_energy2 = 2 * noise_power * sum_an2 + 4 * noise_power * sum_ai_aj
    tau = np.sqrt(noise_energy2 / 2)
    T = (tau * math.sqrt(math.pi / 2) + k * np.sqrt((2 - math.pi / 2) * tau ** 2)) / 1.7
    energy = np.maximum(energy - T[:, None, None], 0.0)
    return (energy.sum(axis=0) + EPS32) / (an.sum(axis=(0, 1)) + EPS32)


def rgb2yiq(x: np.ndarray) -> np.ndarray:
    m = np.array([[0.299, 0.587, 0.114], [0.5959, -0.2746, -0.3213], [0.2115, -0.5227, 0.3112]])
    return np.einsum("kc,chw->khw", m, x)


def gradient_map(lum: np.ndarray) -> np.ndarray:
    """Scharr gradient magnitude, zero padding, cross-correlation (piq gradient_map with scharr_filter and its transpose)."""
    k = np.array([[-3.0, 0.0, 3.0], [-10.0, 0.0, 10.0], [-3.0, 0.0, 3.0]]) / 16.0
    p = np.pad(lum, 1)
    h, w = lum.shape
    gx = np.zeros_like(lum)
    gy = np.zeros_like(lum)
    for a in range(3):
        for b in range(3):
            gx += k[a, b] * p[a:a + h, b:b + w]
            gy += k[b, a] * p[a:a + h, b:b + w]
    return np.sqrt(gx ** 2 + gy ** 2)


def similarity_map(a, b, c):
    return (2.0 * a * b + c) / (a ** 2 + b ** 2 + c)


def avg_pool(x: np.ndarray, k: int) -> np.ndarray:
    c, h, w = x.shape
    ho, wo = h // k, w // k
    return x[:, :ho * k, :wo * k].reshape(c, ho, k, wo, k).mean(axis=(2, 4))


def fsim(x: np.ndarray, y: np.ndarray, data_range: float = 1.0, chromatic: bool = True, fft2=np.fft.fft2, ifft2=np.fft.ifft2) -> float:
    """x, y: [3][H][W] RGB in [0, data_range] -> FSIMc (piq.fsim defaults, one image pair)."""
    x = np.asarray(x, dtype=np.float64) / data_range * 255.0
    y = np.asarray(y, dtype=np.float64) / data_range * 255.0
    k = max(1, round(min(x.shape[-2:]) / 256))
    x, y = avg_pool(x, k), avg_pool(y, k)
    xq, yq = rgb2yiq(x), rgb2yiq(y)
    pcx = phase_congruency(xq[0], fft2=fft2, ifft2=ifft2)
    pcy = phase_congruency(yq[0], fft2=fft2, ifft2=ifft2)
    gmx, gmy = gradient_map(xq[0]), gradient_map(yq[0])
    T1, T2, T3, T4, lmbda = 0.85, 160.0, 200.0, 200.0, 0.03
    pc_max = np.maximum(pcx, pcy)
    score = similarity_map(gmx, gmy, T2) * similarity_map(pcx, pcy, T1) * pc_max
    if chromatic:
        score = score * np.abs(similarity_map(xq[1], yq[1], T3) * similarity_map(xq[2], yq[2], T4)) ** lmbda
    return float(score.sum() / pc_max.sum())


def dft2_explicit(a: np.ndarray, inverse: bool = False) -> np.ndarray:
    """2-D DFT through explicit DFT matrices (the independent evaluation the tests hold numpy's FFT against; also the form the
    device kernels use)."""
    h, w = a.shape
    sgn = 1.0 if inverse else -1.0
    wh = np.exp(sgn * 2j * np.pi * np.outer(np.arange(h), np.arange(h)) / h)
    ww = np.exp(sgn * 2j * np.pi * np.outer(np.arange(w), np.arange(w)) / w)
    out = wh @ a @ ww
    return out / (h * w) if inverse else out
