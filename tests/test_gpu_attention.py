"""The fused attention kernel alone against a float64 restatement of
transformer/Modules.py:14-25, on inputs chosen to exercise every branch: all d_k the config space allows,
ragged key lengths incl. tile-boundary cases, sequence tails, and score spikes that force the online-softmax
reference point to move at a late key tile (the lazy-rescale branch is data dependent and rare)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def ref_attention(qkv, lens, H):
    B, S, d3 = qkv.shape
    d = d3 // 3
    dk = d // H
    x = qkv.double()
    q, k, v = (x[..., i * d:(i + 1) * d].reshape(B, S, H, dk).permute(0, 2, 1, 3) for i in range(3))
    a = q @ k.transpose(-1, -2) / float(np.power(dk, 0.5))
    pad = torch.arange(S)[None, :] >= lens[:, None]
    a = a.masked_fill(pad[:, None, None, :], -np.inf)
    o = torch.softmax(a, dim=-1) @ v
    return o.permute(0, 2, 1, 3).reshape(B, S, d)


@pytest.mark.parametrize("H,dk", [(2, 128), (8, 64), (4, 32)])
@pytest.mark.parametrize("S,lens", [(1, [1]), (33, [33, 1, 32]), (128, [128, 97, 64, 5]), (300, [300, 257, 129])])
def test_attention_vs_float64(H, dk, S, lens):
    from smart_nar_fast_tts_amd import ops

    torch.manual_seed(S * 7 + dk)
    B = len(lens)
    qkv = torch.randn(B, S, 3 * H * dk)
    lens_t = torch.tensor(lens)
    got = ops.attention_core(qkv.cuda(), lens_t.cuda(), H).cpu()
    ref = ref_attention(qkv, lens_t, H)
    for b in range(B):  # padded QUERY rows are computed by the reference too: compare all S rows
        err = (got[b].double() - ref[b]).abs().max().item()
        assert err < 2e-5, (b, err)


@pytest.mark.parametrize("H,dk,S,lens", [(2, 128, 100, [100]), (2, 128, 128, [128, 97, 64, 5] * 4), (2, 128, 788, [788]), (2, 128, 1010, [1010, 700, 33]),
                                         (8, 64, 300, [300, 257, 129]), (4, 32, 130, [130, 1])])
def test_attention_vs_oracle_a5(H, dk, S, lens):
    """Row a5 against the ORACLE's own restatement of ScaledDotProductAttention (oracle/fs2_oracle.py, the function the
    pinned multi_head_attention goes through; fp32 torch-CPU like the reference), on shapes that take each of the three launch
    forms: the strip kernel without a cross-workgroup merge (S <= 128), the strip kernel with the last-arriver merge (single
    utterance, T = 788), and k_attention (config 2's T = 1010).  All S query rows are compared — padded ones are computed by
    the reference too."""
    from oracle import fs2_oracle as orc
    from smart_nar_fast_tts_amd import ops

    torch.manual_seed(S + dk)
    B, d = len(lens), H * dk
    qkv = torch.randn(B, S, 3 * d)
    lens_t = torch.tensor(lens)
    q, k, v = (qkv[..., i * d:(i + 1) * d].reshape(B, S, H, dk).permute(2, 0, 1, 3).reshape(-1, S, dk) for i in range(3))  # SubLayers.py:42-47
    key_pad = torch.arange(S)[None, :] >= lens_t[:, None]
    ref = orc.scaled_dot_product_attention(q, k, v, key_pad.unsqueeze(1).expand(-1, S, -1).repeat(H, 1, 1), dk)
    ref = ref.view(H, B, S, dk).permute(1, 2, 0, 3).reshape(B, S, d)                                                      # SubLayers.py:52-54
    got = ops.attention_core(qkv.cuda(), lens_t.cuda(), H).cpu()
    err = (got - ref).abs().max().item()
    assert err < 2e-5, err


def test_attention_forced_rescale_late_tile():
    """Spike one key per query block far above the rest at a LATE tile so the reference point must move there;
    also a descending spike pattern (first tile largest) so it must NOT move afterwards."""
    from smart_nar_fast_tts_amd import ops

    torch.manual_seed(3)
    B, S, H, dk = 2, 257, 2, 128
    d = H * dk
    qkv = torch.randn(B, S, 3 * d) * 0.5
    q, k = qkv[..., :d], qkv[..., d:2 * d]
    # utterance 0: key 200 (tile 6) aligned with every query of head 0 -> score ~ +60 in natural units
    k[0, 200, :dk] = q[0, :, :dk].mean(0) * 0 + 6.0
    q[0, :, :dk] += 1.0
    # utterance 1: key 3 (tile 0) huge for head 1, later tiles tiny
    k[1, 3, dk:] = 8.0
    q[1, :, dk:] = q[1, :, dk:].abs() + 0.5
    lens_t = torch.tensor([257, 230])
    got = ops.attention_core(qkv.cuda(), lens_t.cuda(), H).cpu()
    ref = ref_attention(qkv, lens_t, H)
    err = (got.double() - ref).abs().max().item()
    assert err < 5e-5, err
    assert torch.isfinite(got).all()


def test_attention_zero_length_gives_nan_like_reference():
    """An utterance with no valid key: softmax over all -inf is NaN in the reference (SURVEY.md §8b Errors)."""
    from smart_nar_fast_tts_amd import ops

    qkv = torch.randn(2, 40, 3 * 256)
    got = ops.attention_core(qkv.cuda(), torch.tensor([0, 40]).cuda(), 2).cpu()
    assert torch.isnan(got[0]).all()
    assert torch.isfinite(got[1]).all()


@pytest.mark.parametrize("H,dk,S,lens", [(2, 128, 1000, [1000]), (2, 128, 700, [700, 130]), (8, 64, 513, [384]), (4, 32, 260, [260, 31, 0])])
def test_attention_split_key_path(H, dk, S, lens):
    """Few workgroups + long key axis (single-utterance latency) takes the split-key path (partials + merge kernel):
    it must agree with float64 and, to rounding, with the single-sweep path on the same input — incl. splits that own
    no valid key tile at all (short utterance beside a long one) and the all-masked NaN case."""
    from smart_nar_fast_tts_amd import ops

    torch.manual_seed(S + dk)
    B = len(lens)
    qkv = torch.randn(B, S, 3 * H * dk)
    qkv[0, S // 2 + 7, H * dk:2 * H * dk] *= 4.0  # a dominant key in a late split: weights differ a lot between partials
    lens_t = torch.tensor(lens)
    split = ops.attention_core(qkv.cuda(), lens_t.cuda(), H, split_scratch=True).cpu()
    single = ops.attention_core(qkv.cuda(), lens_t.cuda(), H, split_scratch=False).cpu()
    ref = ref_attention(qkv, lens_t, H)
    for b in range(B):
        if lens[b] == 0:
            assert torch.isnan(split[b]).all() and torch.isnan(single[b]).all()
            continue
        assert (split[b].double() - ref[b]).abs().max().item() < 2e-5
        assert (split[b] - single[b]).abs().max().item() < 1e-5
