"""Image-space helpers shared by the tests, the fixture generator and bench.py (numpy only)."""
import numpy as np


def block_means(sums, spp, block):
    """Mean linear radiance over the finite pixels of each block x block tile of an f32 sum image (H x W x 3, `spp`
    samples per pixel), and the number of pixels per tile a NaN / infinite sample poisoned."""
    h, w, _ = sums.shape
    finite = np.isfinite(sums).all(axis=2)
    lin = np.where(finite[..., None], sums.astype(np.float64) / spp, 0.0)
    hb, wb = h // block, w // block
    lin = lin[:hb * block, :wb * block].reshape(hb, block, wb, block, 3).sum(axis=(1, 3))
    cnt = finite[:hb * block, :wb * block].reshape(hb, block, wb, block).sum(axis=(1, 3))
    mean = lin / np.maximum(cnt, 1)[..., None]
    return mean.astype(np.float32), (block * block - cnt).astype(np.uint8)


def encode_mean(mean):
    """renderer.rs:241-252 applied to a mean radiance: (255.999 * clamp(sqrt(c), 0, 0.999)) as u8."""
    v = np.sqrt(np.maximum(np.asarray(mean, np.float32), 0))
    v = np.float32(255.999) * np.clip(v, 0, np.float32(0.999))
    return v.astype(np.uint8)


def psnr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else float(10.0 * np.log10(255.0 * 255.0 / mse))


def compare_with_block_fixture(sums, spp, fixture):
    """PSNR (dB, on the encoded 8-bit values) between a rendered frame's block means and a fixture made by
    tools/make_goldens.py (`mean`, `block`, `width`, `height`)."""
    block = int(fixture["block"])
    if sums.shape[0] != int(fixture["height"]) or sums.shape[1] != int(fixture["width"]):
        raise ValueError("frame and fixture differ in size")
    mean, poisoned = block_means(sums, spp, block)
    return {"psnr_db": psnr(encode_mean(mean), encode_mean(fixture["mean"])),
            "blocks": int(mean.shape[0] * mean.shape[1]), "block": block,
            "fixture_spp_per_block": int(fixture["spp"]) * block * block, "frame_spp_per_block": int(spp) * block * block,
            "poisoned_pixels": int(poisoned.sum()), "fixture_poisoned_pixels": int(np.asarray(fixture["poisoned"]).sum()),
            "mean_abs_rel_diff": float(np.mean(np.abs(mean - fixture["mean"]) / np.maximum(fixture["mean"], 1e-3)))}
