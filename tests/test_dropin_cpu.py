"""Structure-driven conversion (implicit_depth_amd.dropin): channel configs are inferred from
the module being replaced and the state_dict transfers 1:1.  CPU only (no kernels run)."""
import torch

import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import cost_volume as cv
from implicit_depth_amd import dropin
from implicit_depth_amd import networks as net


def _same_state(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_conversions_infer_structure_and_copy_weights():
    enc = net.CVEncoder(96, [48, 64, 160, 256], [64, 128, 256, 384])
    syn.fill_state_dict(enc, 1)
    _same_state(enc, dropin.convert_cv_encoder(enc))
    for cls in (net.BDDecoderPP, net.DepthDecoderPP):
        dec = cls([24, 64, 128, 256, 384])
        syn.fill_state_dict(dec, 2)
        new = dropin.convert_decoder(dec)
        assert type(new) is cls
        _same_state(dec, new)
    for cls in (net.SkipDecoder, net.SkipDecoderRegression):
        dec = cls([24, 64, 128, 256, 384])
        syn.fill_state_dict(dec, 5)
        new = dropin.convert_decoder(dec)
        assert type(new) is cls
        _same_state(dec, new)
    for prior in (False, True):
        mlp = net.BinaryMLPNetwork([64, 64, 128, 256], use_prior=prior)
        syn.fill_state_dict(mlp, 3)
        new = dropin.convert_binary_mlp(mlp)
        assert new.use_prior == prior
        _same_state(mlp, new)
    for K in (2, 7):
        fv = cv.FeatureVolumeManager(24, 32, 16, num_source_views=K)
        syn.fill_state_dict(fv.mlp, 4)
        new = dropin.convert_cost_volume(fv)
        assert new.num_source_views == K and new.mlp.net[0].in_features == 26 * K + 20
        _same_state(fv, new)


def test_feature_mlp_column_maps_are_a_permutation_of_the_reference_layout():
    for K in (1, 2, 7):
        vox, pix, pose = cv.feature_mlp_column_maps(K)
        used = [c for c in vox + pix + pose if c >= 0]
        n_in = 16 * (K + 1) + 10 * K + 4
        assert sorted(used) == list(range(n_in)), "every reference MLP input column appears exactly once"
        assert len(vox) == 16 * (K + 4) and len(pix) == 32 and len(pose) == 3 * K
