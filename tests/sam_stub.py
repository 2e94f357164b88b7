"""A deterministic stand-in for SAM's mask decoder (no weights in this image): prompt points -> three mask-logit maps
and three predicted IoUs per point, a pure function of (seed, point coordinates).  tools/make_golden.py feeds it to the
reference's vendored SamAutomaticMaskGenerator through a stub predictor; tests/test_producers.py feeds the same
function to SamAutoMasks -- the two must then produce the same records (tests/golden/sam_generate.npz).

Per (point, mask): an ellipse centred near the point, logit = (1 - rho) * r / s with rho the elliptical radius, so the
mask is the ellipse and the stability score ~ ((r - s)/(r + s))^2; s is drawn so that about half of the masks pass the
0.95 stability cut, the predicted IoUs straddle the 0.88 cut, some masks carry a second, detached blob (boxes are not
just the ellipse's), and the three scales make neighbouring grid points collide in the box NMS."""
import numpy as np


def stub_predict(points_xy: np.ndarray, H: int, W: int, seed: int):
    """points_xy [n, 2] (x, y) in image pixels -> (logits float32 [n, 3, H, W], ious float32 [n, 3])."""
    pts = np.asarray(points_xy, dtype=np.float64)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    logits = np.empty((len(pts), 3, H, W), dtype=np.float32)
    ious = np.empty((len(pts), 3), dtype=np.float32)
    for j, (px, py) in enumerate(pts):
        rng = np.random.Generator(np.random.PCG64([int(seed), int(round(px * 4096.0)), int(round(py * 4096.0))]))
        for m in range(3):
            r = (0.07, 0.15, 0.27)[m] * min(H, W) * rng.uniform(0.6, 1.4)
            asp = rng.uniform(0.6, 1.6)
            cx, cy = px + rng.uniform(-2.0, 2.0), py + rng.uniform(-2.0, 2.0)
            s = r * rng.uniform(0.003, 0.05)
            rho = np.sqrt(((xx - cx) / asp) ** 2 + ((yy - cy) * asp) ** 2) / r
            lg = (1.0 - rho) * (r / s)
            if rng.random() < 0.3:   # a detached second blob
                bx, by, br = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(1.5, 4.0)
                lg = np.maximum(lg, (1.0 - np.sqrt((xx - bx) ** 2 + (yy - by) ** 2) / br) * (br / s))
            logits[j, m] = lg.astype(np.float32)
            ious[j, m] = np.float32(rng.uniform(0.78, 1.0))
    return logits, ious
