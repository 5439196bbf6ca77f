"""Synthetic alert batches of the reference's input shape.

One alert = a 63x63x3 science/reference/difference triplet + 25 metadata scalars
(/root/reference/btsbot/inference_example.py:53-64).  Used by bench.py, smoke() and the tests;
there is no network for real data and the bundled example file (39 alerts) stays in the
reference tree.
"""
import torch

METADATA_COLS = [  # inference_example.py:53-58 == prod_config.json:15-41
    "sgscore1", "distpsnr1", "sgscore2", "distpsnr2", "fwhm", "magpsf",
    "sigmapsf", "chipsf", "ra", "dec", "diffmaglim", "ndethist", "nmtchps",
    "age", "days_since_peak", "days_to_peak", "peakmag_so_far", "new_drb",
    "ncovhist", "nnotdet", "chinr", "sharpnr", "scorr", "sky", "maxmag_so_far"]

# per-column [min, max] measured on the 39 bundled example alerts (rounded outward)
META_RANGES = [
    (0.0, 1.0), (0.0, 12.0), (0.0, 1.0), (0.0, 25.0), (1.0, 6.0), (15.5, 20.0),
    (0.01, 0.25), (1.0, 330.0), (0.0, 360.0), (-30.0, 90.0), (18.0, 21.5), (1.0, 120.0), (0.0, 60.0),
    (0.0, 120.0), (0.0, 100.0), (0.0, 60.0), (15.5, 20.0), (0.0, 1.0),
    (570.0, 745.0), (400.0, 700.0), (0.3, 10.0), (-0.4, 0.4), (5.0, 120.0), (-2.0, 6.0), (16.0, 20.5)]


def synthetic_batch(b: int, seed: int = 2):
    """triplets [b,3,63,63] f32 (|N(0,1)| background + central Gaussian PSF, L2-normalised per
    cutout as alert_utils.py:162-164 does), metadata [b,25] f32 uniform over META_RANGES,
    labels Bernoulli(0.5) int64."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(b, 3, 63, 63, generator=g).abs()
    yy, xx = torch.meshgrid(torch.arange(63.0), torch.arange(63.0), indexing="ij")
    psf = torch.exp(-((yy - 31) ** 2 + (xx - 31) ** 2) / (2 * 1.5 ** 2))
    amp = 20.0 * torch.rand(b, 3, 1, 1, generator=g)
    img = img + amp * psf
    img = img / img.flatten(2).norm(dim=2).reshape(b, 3, 1, 1)
    lo = torch.tensor([r[0] for r in META_RANGES])
    hi = torch.tensor([r[1] for r in META_RANGES])
    meta = lo + (hi - lo) * torch.rand(b, 25, generator=g)
    labels = (torch.rand(b, generator=g) < 0.5).long()
    return img.contiguous(), meta.contiguous(), labels
