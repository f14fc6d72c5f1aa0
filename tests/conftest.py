import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def rel_err(a, ref):
    """Scale-relative error max|a-ref| / max|ref| — the norm SURVEY.md §7 fixes for the
    1e-4 tolerance (element-wise relative error is meaningless near zero crossings)."""
    a = torch.as_tensor(a).double()
    ref = torch.as_tensor(ref).double()
    return ((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def chk(t):
    t = torch.as_tensor(t).double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


TOL = 1e-4  # BASELINE.json: "outputs within 1e-4 rel of reference" (scale-relative)


def block_err(t, blocks_ref, bd, bh, bw):
    """Largest deviation of the (bd, bh, bw)-block sums of a (1, D, H, W) tensor from the reference's (tests/golden/g_full_blocks.npz),
    per block element and relative to the tensor's scale — catches an error confined to a few voxels BETWEEN the slice points of the
    full-size goldens (a single voxel off by 1e-2 of the scale moves its block's figure by 1e-2 / block size)."""
    import numpy as np
    import torch

    t = torch.as_tensor(t).detach().cpu()[0].double()
    D, H, W = t.shape
    b = t.reshape(D // bd, bd, H // bh, bh, W // bw, bw).sum((1, 3, 5)).numpy()
    ref = np.asarray(blocks_ref)
    scale = max(float(t.abs().max()), 1e-30)
    return float(np.abs(b - ref).max()) / (bd * bh * bw) / scale
