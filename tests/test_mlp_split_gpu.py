"""Split-precision ("f16x3") MLP feature volume (csrc/feature_volume.hip fv_mlp_f16_k) against the
reference goldens (G2), the fp64 oracle and the fp32-MFMA kernel — same tolerances as
tests/test_feature_volume_gpu.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import cost_volume as cvmod

import test_feature_volume_gpu as base


@pytest.fixture()
def f16_mlp():
    old = cvmod.DEFAULT_MLP_MATH
    cvmod.DEFAULT_MLP_MATH = "f16x3"
    yield
    cvmod.DEFAULT_MLP_MATH = old


@pytest.mark.parametrize("name", ["g2_small", "g2_b2", "g2_k2"])
def test_matches_reference_golden_f16x3(name, f16_mlp):
    base.test_matches_reference_golden(name)


@pytest.mark.parametrize("shape", [(1, 1, 9, 13, 3), (2, 5, 17, 23, 6), (1, 7, 24, 32, 64), (3, 4, 8, 8, 2), (1, 8, 12, 20, 5)])
def test_matches_oracle_fp64_f16x3(shape, f16_mlp):
    base.test_matches_oracle_fp64(shape)


def test_full_size_golden_f16x3(f16_mlp):
    base.test_full_size_golden()


def test_pipeline_with_feature_volume_f16x3(f16_mlp):
    base.test_pipeline_with_feature_volume()


@pytest.mark.parametrize("K", [7, 8])
def test_f16x3_vs_fp32_kernel_with_scaled_inputs(K):
    """Same volume from both kernels; feature magnitudes of 2^-30 / 2^20 exercise the per-voxel
    power-of-two scaling (the fp32 kernel needs none)."""
    dev = torch.device("cuda:0")
    B, H, W, D = 2, 24, 32, 16
    for gain in (1.0, 2.0 ** -30, 2.0 ** 20):
        d = {k: v.to(dev) for k, v in syn.cost_volume_inputs(B, K, 16, H, W, seed=5).items()}
        d["cur_feats"] = d["cur_feats"] * gain
        d["src_feats"] = d["src_feats"] * gain
        vols = {}
        for math in ("fp32", "f16x3"):
            m = cvmod.FeatureVolumeManager(H, W, D, num_source_views=K).to(dev)
            syn.fill_state_dict(m.mlp, seed=99, gain=1.4)
            m.mlp_math = math
            vol, lowest, planes, mask = m(**d, return_mask=True)
            vols[math] = vol
        a, b = vols["fp32"], vols["f16x3"]
        assert float((a - b).abs().max() / a.abs().max()) < 1e-5, gain


# ---- BinaryMLP (csrc/mlp.hip binary_mlp_k<1, true>) ------------------------------------------------
import test_mlp_gpu as mlp_base


@pytest.mark.parametrize("use_prior", [False, True])
def test_binary_mlp_golden_f16x3(use_prior, f16_mlp):
    mlp_base.test_fused_logits_golden(use_prior)


@pytest.mark.parametrize("use_prior", [False, True])
def test_binary_mlp_module_interface_f16x3(use_prior, f16_mlp):
    mlp_base.test_module_interface_matches_oracle(use_prior)


def test_binary_mlp_odd_sizes_and_scales_f16x3(f16_mlp):
    mlp_base.test_prior_absent_is_minus_one_and_odd_sizes()
    mlp_base.test_all_scales_interface()


def test_binary_depth_search_f16x3(f16_mlp):
    mlp_base.test_fused_binary_depth_search_matches_reference_loop()
    mlp_base.test_fused_search_with_per_depth_thresholder()
