"""Temporal workload (BASELINE.json configs[4]; reference loop inference/inference.py:139-157) with S independent sequences
per GPU in one batch, each carrying its own prior: sequences must not leak into each other and must match the one-sequence
loop."""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(S):
    return argparse.Namespace(batch=S, sequences=S, views=2, planes=16, height=96, width=128, volume="mlp", conv_math="fp32", mlp_math="fp32")


def _frames(wl, n):
    outs = []
    with torch.inference_mode():
        for _ in range(n):
            wl.step()
            outs.append(wl.out["pred_0"].clone())
    torch.cuda.synchronize()
    return outs


def test_sequences_in_one_batch_are_independent_and_match_the_single_sequence_loop():
    from bench import TemporalWorkload

    S, T = 4, 6
    dev = torch.device("cuda:0")
    big = TemporalWorkload(_args(S), dev, 0)
    ref = _frames(big, T)
    assert ref[0].shape[0] == S and big.prev[0].shape[0] == S

    # (a) bit-for-bit independence: the same plan run on the sequences in another order gives the permuted outputs exactly
    perm = torch.tensor([2, 0, 3, 1], device=dev)
    shuf = TemporalWorkload(_args(S), dev, 0)
    shuf.model = big.model  # same weights, same plan
    shuf.d = {k: (v[perm].contiguous() if v.dim() > 0 and v.shape[0] == S else v) for k, v in big.d.items()}
    shuf.pyr = [t[perm].contiguous() for t in big.pyr]
    shuf.l1 = big.l1[perm].contiguous()
    shuf.poses = [(a[perm].contiguous(), b[perm].contiguous()) for a, b in big.poses]
    got = _frames(shuf, T)
    for t in range(T):
        assert torch.equal(got[t], ref[t][perm]), f"frame {t}: sequences of one batch influence each other"

    # (b) each sequence equals the reference-style loop over that sequence alone.  Not bit-for-bit: a one-frame batch takes other
    # tile shapes / kernels (direct instead of Winograd convs, split-K on the small maps), i.e. another summation order — the same
    # bar as test_batch_invariance_at_bench_size, held over the whole carried-prior sequence
    for q in (0, 3):
        one = TemporalWorkload(_args(1), dev, 0)
        one.model.load_state_dict(big.model.state_dict())
        one.d = {k: (v[q:q + 1].contiguous() if v.dim() > 0 and v.shape[0] == S else v) for k, v in big.d.items()}
        one.pyr = [t[q:q + 1].contiguous() for t in big.pyr]
        one.l1 = big.l1[q:q + 1].contiguous()
        one.poses = [(a[q:q + 1].contiguous(), b[q:q + 1].contiguous()) for a, b in big.poses]
        for t, o in enumerate(_frames(one, T)):
            r = ref[t][q:q + 1]
            assert float((o - r).abs().max() / r.abs().max()) < 5e-5, (q, t)


def test_frames_in_flight_chain_matches_the_frame_by_frame_loop():
    """HotPath.forward(frame_chain=...): F consecutive frames of ONE sequence share the head / volume / conv launches, the occlusion MLP and its
    carried prior stay frame by frame.  Must equal the single-frame loop (same bar as the batched-sequences test: the two-frame plan takes other
    tile shapes, i.e. another summation order) over several steps, including the prior handed from one step to the next."""
    from bench import TemporalWorkload

    F, T = 2, 4
    dev = torch.device("cuda:0")
    a = _args(1)
    a.frames_in_flight = F
    big = TemporalWorkload(a, dev, 0)
    assert big.B == F
    one = TemporalWorkload(_args(1), dev, 0)
    one.model.load_state_dict(big.model.state_dict())
    one.poses = big.poses
    with torch.inference_mode():
        for step in range(T):
            big.step()
            got = big.out["pred_0"].clone()
            assert got.shape[0] == F
            for f in range(F):  # the same frames one at a time: frame index 2 * step + f, inputs = batch entry f
                one.d = {k: (v[f:f + 1].contiguous() if v.dim() > 0 and v.shape[0] == F else v) for k, v in big.d.items()}
                one.pyr = [t[f:f + 1].contiguous() for t in big.pyr]
                one.l1 = big.l1[f:f + 1].contiguous()
                assert one.t == F * step + f
                one.step()
                r = one.out["pred_0"]
                assert float((got[f:f + 1] - r).abs().max() / r.abs().max()) < 5e-5, (step, f)
            if step > 0:
                assert "prior_mask" in big.out and big.out["prior_mask"].shape[0] == F
    torch.cuda.synchronize()
