"""Pins oracle/fs2_oracle.py against fixtures generated from the imported reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import fs2_oracle as orc
from tests.util import load_golden, weights_for

TOL = 2e-5  # oracle vs reference: same torch CPU kernels, differences only from op grouping


def _run(meta, z, texts_key="texts"):
    cfg, sd = weights_for(meta)
    w = orc.to_torch_weights(sd)
    stages = {}
    with torch.no_grad():
        out = orc.forward(w, cfg, torch.from_numpy(z["speakers"]), torch.from_numpy(z[texts_key]),
                          torch.from_numpy(z["in_src_lens"]), int(meta["L"]), stages=stages)
    return cfg, w, out, stages


NAMES = ["output", "postnet_output", "p_predictions", "e_predictions", "log_d_predictions", "d_rounded",
         "src_masks", "mel_masks", "src_lens", "mel_lens"]


@pytest.mark.parametrize("name", ["e2e_tiny_single", "e2e_tiny_padded_src", "e2e_tiny_equal_len", "e2e_full_padded_src"])
def test_e2e_12tuple(name):
    meta, z = load_golden(name)
    cfg, w, out, stages = _run(meta, z)
    assert out[10] is None and out[11] is None
    for i, n in enumerate(NAMES):
        got = out[i].numpy()
        ref = z[n]
        assert got.shape == ref.shape and got.dtype == ref.dtype, (n, got.shape, ref.shape, got.dtype, ref.dtype)
        if got.dtype == np.float32:
            assert np.abs(got - ref).max() <= TOL, (n, np.abs(got - ref).max())
        else:
            assert np.array_equal(got, ref), n
    # integer-valued outputs are exact
    assert np.array_equal(out[5].numpy(), z["d_rounded"])
    assert np.array_equal(out[9].numpy(), z["mel_lens"])


@pytest.mark.parametrize("name", ["e2e_tiny_single", "e2e_tiny_padded_src"])
def test_per_module_intermediates(name):
    """(i) per-op goldens captured with forward hooks on the reference's modules."""
    meta, z = load_golden(name)
    cfg, sd = weights_for(meta)
    w = orc.to_torch_weights(sd)
    t = cfg["transformer"]
    cap = {k[4:]: z[k] for k in z.files if k.startswith("cap.")}
    src_mask = torch.from_numpy(z["src_masks"])
    mel_mask = torch.from_numpy(z["mel_masks"])

    def chk(got, ref, what):
        assert got.shape == ref.shape, (what, got.shape, ref.shape)
        assert np.abs(got.numpy() - ref).max() <= TOL, (what, np.abs(got.numpy() - ref).max())

    with torch.no_grad():
        x = torch.from_numpy(cap["enc0_attn.in0"])
        chk(orc.multi_head_attention(w, "txt_encoder.layer_stack.0.slf_attn", x, src_mask, t["encoder_head"]),
            cap["enc0_attn.out0"], "enc attn")
        chk(orc.positionwise_ffn(w, "txt_encoder.layer_stack.0.pos_ffn", torch.from_numpy(cap["enc0_ffn.in0"])),
            cap["enc0_ffn.out0"], "enc ffn")
        x = torch.from_numpy(cap["dec0_attn.in0"])
        chk(orc.multi_head_attention(w, "mel_decoder.layer_stack.0.slf_attn", x, mel_mask, t["decoder_head"]),
            cap["dec0_attn.out0"], "dec attn")
        chk(orc.positionwise_ffn(w, "mel_decoder.layer_stack.0.pos_ffn", torch.from_numpy(cap["dec0_ffn.in0"])),
            cap["dec0_ffn.out0"], "dec ffn")
        chk(orc.txt_encoder(w, torch.from_numpy(z["texts"]), src_mask, t["encoder_head"], cfg["max_seq_len"]),
            cap["enc.out0"], "txt_encoder")
        chk(orc.variance_predictor(w, "variance_adaptor.duration_predictor", torch.from_numpy(cap["dur_pred.in0"]), src_mask),
            cap["dur_pred.out0"], "duration predictor")
        chk(orc.variance_predictor(w, "variance_adaptor.pitch_predictor", torch.from_numpy(cap["pitch_pred.in0"]), mel_mask),
            cap["pitch_pred.out0"], "pitch predictor")
        chk(orc.variance_predictor(w, "variance_adaptor.energy_predictor", torch.from_numpy(cap["energy_pred.in0"]), mel_mask),
            cap["energy_pred.out0"], "energy predictor")
        lr_out, lr_len = orc.length_regulate(torch.from_numpy(cap["lr.in0"]), torch.from_numpy(cap["lr.in1"]), None)
        assert np.array_equal(lr_out.numpy(), cap["lr.out0"]) and np.array_equal(lr_len.numpy(), cap["lr.out1"])
        chk(orc.mel_decoder(w, torch.from_numpy(cap["dec.in0"]), mel_mask, t["decoder_head"], cfg["max_seq_len"]),
            cap["dec.out0"], "mel_decoder")
        chk(torch.nn.functional.linear(torch.from_numpy(cap["mel_linear.in0"]), w["mel_linear.weight"], w["mel_linear.bias"]),
            cap["mel_linear.out0"], "mel_linear")
        chk(orc.postnet(w, torch.from_numpy(cap["postnet.in0"])), cap["postnet.out0"], "postnet")
        # a11: the energy predictor's input is x + pitch_embedding[bucketize(pitch)] with NO mask
        p_pred, p_emb = orc.variance_embedding(w, "pitch", torch.from_numpy(cap["pitch_pred.in0"]), mel_mask, 1.0)
        chk(torch.from_numpy(cap["pitch_pred.in0"]) + p_emb, cap["energy_pred.in0"], "pitch embedding add")


def test_targets_branch():
    """p_targets / e_targets handed to forward() in the inference branch (model/modules.py:82-84,93-95)."""
    meta, z = load_golden("e2e_tiny_targets")
    cfg, sd = weights_for(meta)
    with torch.no_grad():
        out = orc.forward(orc.to_torch_weights(sd), cfg, torch.from_numpy(z["speakers"]), torch.from_numpy(z["texts"]),
                          torch.from_numpy(z["in_src_lens"]), int(meta["L"]),
                          p_targets=torch.from_numpy(z["p_targets"]), e_targets=torch.from_numpy(z["e_targets"]))
    for i, n in enumerate(NAMES):
        got, ref = out[i].numpy(), z[n]
        assert got.shape == ref.shape
        if got.dtype == np.float32:
            assert np.abs(got - ref).max() <= TOL, n
        else:
            assert np.array_equal(got, ref), n


def test_neighbours_bit_identical():
    """F3 case (c): same utterance, two different longer neighbours → identical results."""
    meta, z = load_golden("e2e_tiny_neighbours")
    assert meta["reference_bit_identical"]
    cfg, sd = weights_for(meta)
    w = orc.to_torch_weights(sd)
    outs = {}
    for tag in ("A", "B"):
        with torch.no_grad():
            outs[tag] = orc.forward(w, cfg, torch.from_numpy(z["speakers"]), torch.from_numpy(z["texts" + tag]),
                                    torch.from_numpy(z["in_src_lens"]), int(meta["L"]))
        for i, n in enumerate(NAMES):
            got, ref = outs[tag][i].numpy(), z[f"{tag}.{n}"]
            assert got.shape == ref.shape
            if got.dtype == np.float32:
                assert np.abs(got - ref).max() <= TOL, (tag, n)
            else:
                assert np.array_equal(got, ref)
    n = int(z["A.mel_lens"][1])
    assert np.array_equal(outs["A"][1].numpy()[1, :n], outs["B"][1].numpy()[1, :n])


@pytest.mark.parametrize("name", ["e2e_tiny_T_below_1000", "e2e_tiny_T_above_1000"])
def test_position_table_switch(name):
    meta, z = load_golden(name)
    cfg, w, out, _ = _run(meta, z)
    assert np.array_equal(out[9].numpy(), z["mel_lens"])
    assert np.array_equal(out[5].numpy(), z["d_rounded"])
    assert np.abs(out[1].numpy() - z["postnet_output"]).max() <= TOL
    assert np.abs(out[2].numpy() - z["p_predictions"]).max() <= 2e-3  # pitch is O(500)
    assert np.abs(out[3].numpy() - z["e_predictions"]).max() <= 1e-4


def test_kat_integer():
    meta, z = load_golden("kat_integer")
    # a9 incl. exact halves (round-half-even), -0.0 survival and negative-before-clamp
    dr = orc.duration_round(torch.from_numpy(z["logd"])).numpy()
    assert np.array_equal(dr.view(np.uint32), z["d_rounded"].view(np.uint32))  # bit pattern: -0.0 stays -0.0
    # a10
    o, l = orc.length_regulate(torch.from_numpy(z["lr_x"]), torch.from_numpy(z["lr_dur"]), None)
    assert np.array_equal(o.numpy(), z["lr_out"]) and np.array_equal(l.numpy(), z["lr_len"])
    o, l = orc.length_regulate(torch.from_numpy(z["lr_x"]), torch.from_numpy(z["lr_dur"]), 12)
    assert np.array_equal(o.numpy(), z["lr_out_cap12"]) and np.array_equal(l.numpy(), z["lr_len_cap12"])
    # a1
    lens = torch.from_numpy(z["mask_lens"])
    assert np.array_equal(orc.get_mask_from_lengths(lens).numpy(), z["mask_auto"])
    assert np.array_equal(orc.get_mask_from_lengths(lens, 7).numpy(), z["mask_fixed7"])
    # a11
    v = torch.from_numpy(z["bk_vals"])
    assert np.array_equal(orc.bucketize(v, torch.from_numpy(z["pitch_bins"])).numpy(), z["bk_pitch"])
    assert np.array_equal(orc.bucketize(v, torch.from_numpy(z["energy_bins"])).numpy(), z["bk_energy"])
    # a2
    tab = orc.sinusoid_table(4001, 256).numpy()
    assert np.array_equal(tab[z["sin_rows"]], z["sin_tab"])


def test_kat_gaussian_upsampling():
    meta, z = load_golden("kat_gaussian_upsampling")
    out, s, w = orc.gaussian_upsampling(torch.from_numpy(z["x"]), torch.from_numpy(z["d"]), None)
    assert np.abs(out.numpy() - z["out"]).max() <= 1e-6
    assert np.array_equal(s.numpy(), z["s"])
    assert np.abs(w.numpy() - z["w"]).max() <= 1e-7
    out40, _, _ = orc.gaussian_upsampling(torch.from_numpy(z["x"]), torch.from_numpy(z["d"]), 40)
    assert np.abs(out40.numpy() - z["out_maxlen40"]).max() <= 1e-6


@pytest.mark.parametrize("name", ["pin_cfg1_single", "pin_cfg2_b16", "pin_cfg3_b128_sharded", "pin_cfg4_d512",
                                  "pin_cfg5_longform"])
def test_baseline_size_pins(name):
    """BASELINE.json's configs at full size: integer outputs exactly, mel on every 16th frame."""
    meta, z = load_golden(name)
    cfg, w, out, _ = _run(meta, z)
    assert np.array_equal(out[9].numpy(), z["mel_lens"])
    assert np.array_equal(out[5].numpy(), z["d_rounded"])
    st = meta["frame_stride"]
    assert np.abs(out[0].numpy()[:, ::st] - z["output_sub"]).max() <= 1e-4
    assert np.abs(out[1].numpy()[:, ::st] - z["postnet_output_sub"]).max() <= 1e-4


LEVEL_CASES = ["e2e_tiny_phoneme_level", "e2e_tiny_pitch_phoneme_energy_frame", "e2e_tiny_pitch_frame_energy_phoneme",
               "e2e_tiny_gaussian_wired"]


@pytest.mark.parametrize("name", LEVEL_CASES)
def test_feature_levels_and_gaussian_regulator(name):
    """§8 f4 (phoneme_level pitch/energy, model/modules.py:117-126) and §8 f1 (GaussianUpsampling wired in, an extension)."""
    meta, z = load_golden(name)
    cfg, sd = weights_for(meta)
    with torch.no_grad():
        out = orc.forward(orc.to_torch_weights(sd), cfg, torch.from_numpy(z["speakers"]), torch.from_numpy(z["texts"]),
                          torch.from_numpy(z["in_src_lens"]), int(meta["L"]),
                          p_control=meta.get("p_control", 1.0), e_control=meta.get("e_control", 1.0),
                          pitch_level=meta.get("pitch_level", "frame_level"), energy_level=meta.get("energy_level", "frame_level"),
                          length_regulator=meta.get("length_regulator", "hard"))
    for i, n in enumerate(NAMES):
        got, ref = out[i].numpy(), z[n]
        assert got.shape == ref.shape and got.dtype == ref.dtype, (n, got.shape, ref.shape)
        if got.dtype == np.float32:
            assert np.abs(got - ref).max() <= TOL * max(1.0, np.abs(ref).max() / 50), (n, np.abs(got - ref).max())
        else:
            assert np.array_equal(got, ref), n
