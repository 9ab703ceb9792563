"""Parity of the HIP path (through the C-ABI) against (1) the golden fixtures generated from the
imported reference and (2) the oracle on the same seeded inputs.  Needs a real MI355X.

Tolerances (north_star): mel / postnet mel max-abs < 1e-3 (fp32) against the reference CPU
forward; integer frame counts and durations identical.  Per-op checks use 2e-4.
"""
import numpy as np
import pytest
import torch

from tests.util import load_golden, weights_for

pytestmark = pytest.mark.gpu

MEL_TOL = 1e-3   # north_star: mel max-abs error < 1e-3 (fp32)
OP_TOL = 2e-4    # single operators on O(1) activations
NAMES = ["output", "postnet_output", "p_predictions", "e_predictions", "log_d_predictions", "d_rounded",
         "src_masks", "mel_masks", "src_lens", "mel_lens"]

_MODEL = {}


def gpu_model(meta):
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    key = (meta["config"], meta.get("weight_seed", 0), meta["frames_per_phoneme"], meta.get("dur_weight_scale", 0.25))
    if key not in _MODEL:
        _MODEL.clear()
        cfg, sd = weights_for(meta)
        m = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
        m.load_state_dict(sd)
        _MODEL[key] = (cfg, sd, m)
    return _MODEL[key]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_gpu(m, z, meta, texts_key="texts", p_targets=None, e_targets=None):
    with torch.no_grad():
        out = m(dev(z["speakers"]), dev(z[texts_key]), dev(z["in_src_lens"]), int(meta["L"]),
                p_targets=None if p_targets is None else dev(p_targets),
                e_targets=None if e_targets is None else dev(e_targets))
    torch.cuda.synchronize()
    return out


from oracle.parity import EDGE_REL  # noqa: E402  relative distance to a bucket edge below which a flip is fp32 noise (oracle/parity.py)


def bucket_flips(sd, out, z, which="pe", what="free run"):
    """torch.bucketize is discontinuous: where the reference's pitch/energy value sits within fp32 summation noise
    of a bin edge, a different (equally valid) fp32 evaluation may pick the neighbouring embedding row.
    Returns the number of such flips; any flip AWAY from an edge is a failure."""
    n = 0
    valid = ~z["mel_masks"]
    for i, key, bins, edge in ((2, "p_predictions", "variance_adaptor.pitch_bins", "p_edge_rel"),
                               (3, "e_predictions", "variance_adaptor.energy_bins", "e_edge_rel")):
        if key[0] not in which:
            continue
        got = np.searchsorted(sd[bins], out[i].cpu().numpy(), side="left")
        ref = np.searchsorted(sd[bins], z[key], side="left")
        flip = (got != ref) & valid
        assert np.all(z[edge][flip] < EDGE_REL), (what, key, "bucket flip away from a bin edge", z[edge][flip].max())
        assert np.all(np.abs(got - ref)[flip] <= 1), (what, key, "bucket moved by more than one")
        n += int(flip.sum())
    return n


def close(got, ref, tol, what):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    # (absolute, as north_star states it: round 5 added 1e-5 |ref| of slack — 5e-3 on a pitch of 500 — which the chunked accumulation
    # of round 6 no longer needs: worst pitch error 5.5e-4 at |v| ~ 500, worst mel error 3.6e-6)
    assert np.all(err <= tol), (what, "max-abs err", float(err.max()), "tol", tol)
    return float(err.max())


def check_12tuple(out, z, prefix=""):
    assert len(out) == 12 and out[10] is None and out[11] is None
    for i, n in enumerate(NAMES):
        ref = z[prefix + n]
        got = out[i].detach().cpu().numpy()
        assert got.dtype == ref.dtype, (n, got.dtype, ref.dtype)
        assert got.shape == ref.shape, (n, got.shape, ref.shape)
    # integers: identical
    assert np.array_equal(out[9].cpu().numpy(), z[prefix + "mel_lens"]), "frame counts differ"
    assert np.array_equal(out[5].cpu().numpy(), z[prefix + "d_rounded"]), "durations differ"
    assert np.array_equal(out[6].cpu().numpy(), z[prefix + "src_masks"])
    assert np.array_equal(out[7].cpu().numpy(), z[prefix + "mel_masks"])
    assert np.array_equal(out[8].cpu().numpy(), z[prefix + "src_lens"])
    errs = {
        "mel": close(out[0], z[prefix + "output"], MEL_TOL, "mel"),
        "postnet": close(out[1], z[prefix + "postnet_output"], MEL_TOL, "postnet mel"),
        "pitch": close(out[2], z[prefix + "p_predictions"], MEL_TOL, "pitch"),
        "energy": close(out[3], z[prefix + "e_predictions"], MEL_TOL, "energy"),
        "log_d": close(out[4], z[prefix + "log_d_predictions"], 1e-4, "log durations"),
    }
    return errs


@pytest.mark.parametrize("name", ["e2e_tiny_single", "e2e_tiny_padded_src", "e2e_tiny_equal_len", "e2e_full_padded_src"])
def test_e2e_vs_reference_golden(name):
    meta, z = load_golden(name)
    cfg, sd, m = gpu_model(meta)
    errs = check_12tuple(run_gpu(m, z, meta), z)
    print(name, errs)


@pytest.mark.parametrize("name", ["e2e_tiny_single", "e2e_tiny_padded_src"])
def test_per_op_vs_reference_golden(name):
    """(i) every operator of SURVEY.md §8(a) against inputs/outputs hooked from the reference's modules."""
    from smart_nar_fast_tts_amd import ops

    meta, z = load_golden(name)
    cfg, sd, m = gpu_model(meta)
    cap = {k[4:]: z[k] for k in z.files if k.startswith("cap.")}
    src_lens = dev(z["in_src_lens"])
    mel_lens = dev(z["mel_lens"])
    e = {}
    e["enc_attn"] = close(ops.multi_head_attention(m, "txt_encoder.layer_stack.0.slf_attn", dev(cap["enc0_attn.in0"]), src_lens),
                          cap["enc0_attn.out0"], OP_TOL, "a4 enc attention")
    e["enc_ffn"] = close(ops.positionwise_ffn(m, "txt_encoder.layer_stack.0.pos_ffn", dev(cap["enc0_ffn.in0"])),
                         cap["enc0_ffn.out0"], OP_TOL, "a6 enc ffn")
    e["dec_attn"] = close(ops.multi_head_attention(m, "mel_decoder.layer_stack.0.slf_attn", dev(cap["dec0_attn.in0"]), mel_lens),
                          cap["dec0_attn.out0"], OP_TOL, "a4 dec attention")
    e["dec_ffn"] = close(ops.positionwise_ffn(m, "mel_decoder.layer_stack.0.pos_ffn", dev(cap["dec0_ffn.in0"])),
                         cap["dec0_ffn.out0"], OP_TOL, "a6 dec ffn")
    e["txt_encoder"] = close(ops.txt_encoder(m, dev(z["texts"]), src_lens), cap["enc.out0"], OP_TOL, "a3 txt_encoder")
    # a7 FFTBlock.forward (transformer/Layers.py:39-48) called directly: these fixtures are 1+1-layer models, so layer 0's
    # input is the attention hook's input and its output (both masked_fill's applied) is the stack's output
    assert cfg["transformer"]["encoder_layer"] == 1 and cfg["transformer"]["decoder_layer"] == 1
    e["enc_fft_block"] = close(ops.fft_block(m, "txt_encoder.layer_stack.0", dev(cap["enc0_attn.in0"]), src_lens),
                               cap["enc.out0"], OP_TOL, "a7 enc FFTBlock")
    e["dec_fft_block"] = close(ops.fft_block(m, "mel_decoder.layer_stack.0", dev(cap["dec0_attn.in0"]), mel_lens),
                               cap["dec.out0"], OP_TOL, "a7 dec FFTBlock")
    e["dur_pred"] = close(ops.variance_predictor(m, "variance_adaptor.duration_predictor", dev(cap["dur_pred.in0"]), src_lens),
                          cap["dur_pred.out0"], OP_TOL, "a8 duration predictor")
    e["pitch_pred"] = close(ops.variance_predictor(m, "variance_adaptor.pitch_predictor", dev(cap["pitch_pred.in0"]), mel_lens),
                            cap["pitch_pred.out0"], 1e-3, "a8 pitch predictor")
    e["energy_pred"] = close(ops.variance_predictor(m, "variance_adaptor.energy_predictor", dev(cap["energy_pred.in0"]), mel_lens),
                             cap["energy_pred.out0"], OP_TOL, "a8 energy predictor")
    lr_out, lr_len = ops.length_regulate(dev(cap["lr.in0"]), dev(cap["lr.in1"]))
    assert np.array_equal(lr_out.cpu().numpy(), cap["lr.out0"]), "a10 length regulator must be bit-exact (it is a gather)"
    assert np.array_equal(lr_len.cpu().numpy(), cap["lr.out1"])
    p_pred, x_p = ops.variance_embedding(m, "pitch", dev(cap["pitch_pred.in0"]), mel_lens, 1.0)
    e["pitch_embed_add"] = close(x_p, cap["energy_pred.in0"], OP_TOL, "a11 x + pitch_embedding (unmasked)")
    e["mel_decoder"] = close(ops.mel_decoder(m, dev(cap["dec.in0"]), mel_lens), cap["dec.out0"], OP_TOL, "a13 mel_decoder")
    e["mel_linear"] = close(ops.mel_linear(m, dev(cap["mel_linear.in0"])), cap["mel_linear.out0"], OP_TOL, "a14 mel_linear")
    e["postnet"] = close(ops.postnet(m, dev(cap["postnet.in0"])), cap["postnet.out0"], OP_TOL, "a15 postnet")
    print(name, {k: f"{v:.2e}" for k, v in e.items()})


def test_neighbours_padding_invariance():
    """the same utterance beside two different longer neighbours (different T_pad).
    The reference is bit-identical there; the HIP path must match each run and agree with itself."""
    meta, z = load_golden("e2e_tiny_neighbours")
    cfg, sd, m = gpu_model(meta)
    outA = run_gpu(m, z, meta, "textsA")
    outB = run_gpu(m, z, meta, "textsB")
    check_12tuple(outA, z, "A.")
    check_12tuple(outB, z, "B.")
    n = int(z["A.mel_lens"][1])
    a, b = outA[1][1, :n].cpu().numpy(), outB[1][1, :n].cpu().numpy()
    assert np.array_equal(a, b), ("utterance result depends on its neighbour", float(np.abs(a - b).max()))


@pytest.mark.parametrize("name", ["e2e_tiny_T_below_1000", "e2e_tiny_T_above_1000"])
def test_position_table_switch(name):
    """Cached sinusoid table (T <= max_seq_len) vs on-device rebuild (T > max_seq_len), transformer/Models.py:218-235."""
    meta, z = load_golden(name)
    cfg, sd, m = gpu_model(meta)
    out = run_gpu(m, z, meta)
    assert np.array_equal(out[9].cpu().numpy(), z["mel_lens"])
    assert np.array_equal(out[5].cpu().numpy(), z["d_rounded"])
    close(out[2], z["p_predictions"], 2e-3, "pitch")
    from oracle import parity
    valid = ~z["mel_masks"]
    dp = parity.max_rel_deviation(out[2].cpu().numpy(), z["p_predictions"], sd["variance_adaptor.pitch_bins"], valid)
    assert dp <= EDGE_REL, ("pitch deviates from the reference by more than the recorded noise bound", dp)
    n_flip = bucket_flips(sd, out, z, "p")
    tf = run_gpu(m, z, meta, p_targets=z["p_predictions"], e_targets=z["e_predictions"])
    close(tf[3], z["e_predictions"], MEL_TOL, "energy")
    de = parity.max_rel_deviation(tf[3].cpu().numpy(), z["e_predictions"], sd["variance_adaptor.energy_bins"], valid)
    assert de <= EDGE_REL, ("energy deviates from the reference by more than the recorded noise bound", de)
    n_flip += bucket_flips(sd, tf, z, "e", "pitch-pinned run")
    e = close(tf[1], z["postnet_output"], MEL_TOL, "postnet mel")
    print(name, "T", out[0].shape[1], "edge bucket flips", n_flip, "postnet err", e)


def test_targets_branch_vs_reference_golden():
    """forward(p_targets=, e_targets=) in the inference branch (model/fastspeech2_align.py:70-78, model/modules.py:82-84,93-95)."""
    meta, z = load_golden("e2e_tiny_targets")
    cfg, sd, m = gpu_model(meta)
    out = run_gpu(m, z, meta, p_targets=z["p_targets"], e_targets=z["e_targets"])
    print(check_12tuple(out, z))


def test_kat_integer_ops():
    from smart_nar_fast_tts_amd import ops

    meta, z = load_golden("kat_integer")
    # a9: exact halves round to even, -0.0 survives the clamp, negatives clamp to 0
    dr = ops.duration_round(dev(z["logd"])).cpu().numpy()
    assert np.array_equal(dr.view(np.uint32), z["d_rounded"].view(np.uint32)), (dr, z["d_rounded"])
    # a10: -0.0 / 0 / negative / fractional durations; default and explicit max_len
    o, l = ops.length_regulate(dev(z["lr_x"]), dev(z["lr_dur"]))
    assert np.array_equal(o.cpu().numpy(), z["lr_out"]) and np.array_equal(l.cpu().numpy(), z["lr_len"])
    o, l = ops.length_regulate(dev(z["lr_x"]), dev(z["lr_dur"]), 12)
    assert np.array_equal(o.cpu().numpy(), z["lr_out_cap12"]) and np.array_equal(l.cpu().numpy(), z["lr_len_cap12"])
    # a1
    lens = dev(z["mask_lens"])
    assert np.array_equal(ops.mask_from_lengths(lens).cpu().numpy(), z["mask_auto"])
    assert np.array_equal(ops.mask_from_lengths(lens, 7).cpu().numpy(), z["mask_fixed7"])
    # a11: at / just below / just above edges, below bins[0], above bins[-1]
    v = dev(z["bk_vals"])
    assert np.array_equal(ops.bucketize(v, dev(z["pitch_bins"])).cpu().numpy(), z["bk_pitch"])
    assert np.array_equal(ops.bucketize(v, dev(z["energy_bins"])).cpu().numpy(), z["bk_energy"])
    # a2: rows 0,1,2,999,1000,1001,3999,4000 of the float64-evaluated table
    tab = ops.sinusoid_table(4001, 256).cpu().numpy()
    err = np.abs(tab[z["sin_rows"]] - z["sin_tab"]).max()
    assert err <= 1.2e-7, err  # one float32 ulp of values in [-1,1]: device libm vs numpy in float64 before the cast


def test_kat_gaussian_upsampling():
    """a12: built as a standalone kernel (dead code in the reference forward)."""
    from smart_nar_fast_tts_amd import ops

    meta, z = load_golden("kat_gaussian_upsampling")
    out, s, w = ops.gaussian_upsampling(dev(z["x"]), dev(z["d"]))
    close(out, z["out"], 5e-6, "gaussian out")
    assert np.array_equal(s.cpu().numpy(), z["s"])
    close(w, z["w"], 1e-6, "gaussian w")
    out40, _, _ = ops.gaussian_upsampling(dev(z["x"]), dev(z["d"]), 40)
    close(out40, z["out_maxlen40"], 5e-6, "gaussian out (max_len=40)")


@pytest.mark.parametrize("B,L,D", [(2, 300, 256), (1, 513, 512), (3, 7, 256), (2, 257, 64), (2, 40, 24), (1, 9, 600)])
def test_gaussian_upsampling_scales_with_L(B, L, D):
    """a12 beyond the reference KAT's sizes: phoneme axes longer than one LDS chunk of weights (256), odd lengths (the
    MFMA walks phonemes two at a time), 64..512 channels, zero and fractional durations — against the oracle's
    GaussianUpsampling (model/modules.py:166-192) on the same inputs."""
    from oracle import fs2_oracle as orc
    from smart_nar_fast_tts_amd import ops

    rs = np.random.RandomState(L)
    x = rs.standard_normal((B, L, D)).astype(np.float32)
    d = np.maximum(rs.randint(-1, 6, size=(B, L)).astype(np.float32) + rs.choice([0.0, 0.5], size=(B, L)).astype(np.float32), 0.0)
    ref_out, ref_s, ref_w = orc.gaussian_upsampling(torch.from_numpy(x), torch.from_numpy(d), None)
    out, s_, w = ops.gaussian_upsampling(dev(x), dev(d))
    assert np.array_equal(s_.cpu().numpy().reshape(-1), ref_s.numpy().reshape(-1))
    e_w = close(w, ref_w.numpy(), 1e-6, "gaussian w")
    e_o = close(out, ref_out.numpy(), 2e-5, "gaussian out")
    T = ref_out.shape[1]
    padded, _, _ = ops.gaussian_upsampling(dev(x), dev(d), T + 37)
    assert padded.shape[1] == T + 37 and torch.equal(padded[:, :T], out) and float(padded[:, T:].abs().max()) == 0.0
    print(f"gaussian B={B} L={L} D={D} T={T}: w err {e_w:.1e} out err {e_o:.1e}")


@pytest.mark.parametrize("name", ["pin_cfg1_single", "pin_cfg2_b16", "pin_cfg3_b128_sharded", "pin_cfg4_d512",
                                  "pin_cfg5_longform"])
def test_baseline_configs_vs_reference_pins(name):
    """BASELINE.json's configs against numbers produced by the reference itself, at the sizes the committed fixtures
    cover: configs 1 and 2 at FULL size (B=1 L=100; B=16 L=128), config 3 as ONE of its eight 16-utterance shards,
    config 4 (d=512, 6+6 layers, 8 heads) at B=8 of its 64, config 5 (T~3900) at B=2 of its 8 — the reference needs
    minutes and tens of GB for the full B=64 / B=8 shapes; those run against the oracle in
    test_full_size_cfg4_cfg5_vs_oracle below.  The batch size in use is printed.
    Free run: every duration and frame count exactly; pitch/energy within tolerance; bucket choices identical
    except where the reference value sits on a bin edge to within fp32 noise (counted and printed).
    Then, with the reference's own pitch/energy values handed in as p_targets/e_targets (so both sides take the
    same discrete decisions), mel / postnet mel on every 16th frame within 1e-3."""
    meta, z = load_golden(name)
    cfg, sd, m = gpu_model(meta)
    out = run_gpu(m, z, meta)
    flips = np.flatnonzero(out[5].cpu().numpy().ravel() != z["d_rounded"].ravel())
    assert flips.size == 0, ("duration flips at", flips[:8], "half-distance there", z["half_dist"].ravel()[flips[:8]])
    assert np.array_equal(out[9].cpu().numpy(), z["mel_lens"])
    assert np.array_equal(out[7].cpu().numpy(), z["mel_masks"])
    close(out[2], z["p_predictions"], 2e-3, "pitch")
    from oracle import parity
    valid = ~z["mel_masks"]
    dp = parity.max_rel_deviation(out[2].cpu().numpy(), z["p_predictions"], sd["variance_adaptor.pitch_bins"], valid)
    assert dp <= EDGE_REL, ("pitch deviates from the reference by more than the recorded noise bound", dp)
    n_flip = bucket_flips(sd, out, z, "p")
    st = meta["frame_stride"]
    # the energy predictor reads x + pitch_embedding: compare it (and everything downstream) with the pitch buckets
    # pinned to the reference's; an edge flip of a pitch bucket legitimately moves energy at the 5 frames around it
    tf = run_gpu(m, z, meta, p_targets=z["p_predictions"], e_targets=z["e_predictions"])
    close(tf[3], z["e_predictions"], MEL_TOL, "energy")
    de = parity.max_rel_deviation(tf[3].cpu().numpy(), z["e_predictions"], sd["variance_adaptor.energy_bins"], valid)
    assert de <= EDGE_REL, ("energy deviates from the reference by more than the recorded noise bound", de)
    n_flip += bucket_flips(sd, tf, z, "e", "pitch-pinned run")
    if n_flip == 0:
        close(out[0][:, ::st], z["output_sub"], MEL_TOL, "mel (free run)")
    e1 = close(tf[0][:, ::st], z["output_sub"], MEL_TOL, "mel")
    e2 = close(tf[1][:, ::st], z["postnet_output_sub"], MEL_TOL, "postnet mel")
    print(name, "B", int(meta["B"]), "L", int(meta["L"]), "T", out[0].shape[1], "frames", int(z["mel_lens"].sum()),
          "edge bucket flips in free run", n_flip, "max in-range relative deviation pitch / energy", dp, de,
          "mel err", e1, "postnet err", e2)


def _half_dist(log_d):
    """distance of exp(log_d) - 1 to the nearest .5 (the rounding boundary of model/modules.py:132-135)"""
    return np.abs((np.exp(np.asarray(log_d, dtype=np.float64)) - 1.0) % 1.0 - 0.5)


def _oracle_case_with_margin(w, cfg, B, L, seed, lens=None, margin=2e-4, tries=40, **kw):
    """First input seed >= `seed` whose ORACLE durations all sit >= margin away from a rounding boundary, so that an
    fp32 summation-order difference cannot flip one: the comparison below then always runs (no silent skip)."""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import fs2_oracle as orc

    for s in range(seed, seed + tries):
        inp = wl.synth_inputs(B, L, seed=s, src_lens=lens)
        with torch.no_grad():
            ref = orc.forward(w, cfg, torch.from_numpy(inp[0]), torch.from_numpy(inp[1]), torch.from_numpy(inp[2]), inp[3], **kw)
        valid = ~ref[6].numpy()
        if not valid.any() or _half_dist(ref[4].numpy())[valid].min() >= margin:
            return s, inp, ref
    raise AssertionError(f"no input seed in [{seed}, {seed + tries}) keeps every duration {margin} away from a rounding boundary")


def _screened_full_batch(w, cfg, B, L, seed, margin=2e-4, lens_fn=None, **oracle_kw):
    """Full-size configs: the oracle forward takes 2-20 s there, so instead of re-drawing whole batches, screen 2*B candidate
    utterances with the oracle's encoder + duration predictor only (7 % of the flops; an unpadded utterance's durations
    depend on its own tokens alone) and keep the first B whose durations all sit >= margin away from a
    rounding boundary.  Returns the batch and the oracle's full forward on it."""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import fs2_oracle as orc

    # lens_fn(n) -> n phoneme counts (ragged batches); a padded utterance's durations still depend on its own tokens and
    # on L alone (key mask + zeroed pad rows), so the per-utterance screen stays valid
    sp, tx, ln, Lmax = wl.synth_inputs(2 * B, L, seed=seed, src_lens=None if lens_fn is None else lens_fn(2 * B))
    t = cfg["transformer"]
    keep = []
    with torch.no_grad():
        for lo in range(0, 2 * B, 16):
            txc, lnc = torch.from_numpy(tx[lo:lo + 16]), torch.from_numpy(ln[lo:lo + 16])
            mask = orc.get_mask_from_lengths(lnc, Lmax)
            x = orc.txt_encoder(w, txc, mask, t["encoder_head"], cfg["max_seq_len"])
            hd = _half_dist(orc.variance_predictor(w, "variance_adaptor.duration_predictor", x, mask).numpy())
            keep += [lo + i for i in range(hd.shape[0]) if hd[i][:int(lnc[i])].min() >= margin]
    assert len(keep) >= B, f"only {len(keep)} of {2 * B} candidate utterances keep {margin} from the rounding boundaries"
    keep = keep[:B]
    inp = (sp[keep], np.ascontiguousarray(tx[keep]), ln[keep], Lmax)
    with torch.no_grad():
        ref = orc.forward(w, cfg, torch.from_numpy(inp[0]), torch.from_numpy(inp[1]), torch.from_numpy(inp[2]), Lmax, **oracle_kw)
    assert _half_dist(ref[4].numpy())[~ref[6].numpy()].min() >= margin / 2
    return inp, ref


def _pinned_vs_oracle(m, w, cfg, inp, ref, what, **kw):
    """Durations / frame counts / masks identical; pitch within 2e-3 with only at-edge bucket flips; then energy, mel and
    PostNet mel on EVERY frame with both sides taking the same bucket decisions (the oracle's values as targets)."""
    from oracle import fs2_oracle as orc
    from oracle import parity

    with torch.no_grad():
        out = m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3], **kw)
    assert np.array_equal(out[5].cpu().numpy(), ref[5].numpy()), (what, "durations differ")
    assert np.array_equal(out[9].cpu().numpy(), ref[9].numpy()), (what, "frame counts differ")
    assert np.array_equal(out[7].cpu().numpy(), ref[7].numpy()), (what, "mel masks differ")
    close(out[4], ref[4].numpy(), 1e-4, what + " log durations")
    close(out[2], ref[2].numpy(), 2e-3, what + " pitch")
    valid = ~ref[7].numpy()
    fp = parity.classify_bucket_flips(out[2].cpu().numpy(), ref[2].numpy(), w["variance_adaptor.pitch_bins"].numpy(), valid)
    assert fp[1] == 0 and fp[2] == 0, (what, "pitch bucket flips away from a bin edge / by more than one", fp)
    dp = parity.max_rel_deviation(out[2].cpu().numpy(), ref[2].numpy(), w["variance_adaptor.pitch_bins"].numpy(), valid)
    assert dp <= parity.EDGE_REL, (what, "pitch deviates from the oracle by more than the recorded noise bound", dp)
    pc, ec = kw.get("p_control", 1.0), kw.get("e_control", 1.0)
    with torch.no_grad():
        # the oracle's free-run predictions are already scaled by p/e_control; as targets they are bucketized as is
        tf = m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3], p_targets=ref[2].cuda(), e_targets=ref[3].cuda())
        pp = m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3], p_targets=ref[2].cuda(), e_control=ec)
    fe = parity.classify_bucket_flips(pp[3].cpu().numpy(), ref[3].numpy(), w["variance_adaptor.energy_bins"].numpy(), valid)
    assert fe[1] == 0 and fe[2] == 0, (what, "energy bucket flips away from a bin edge / by more than one", fe)
    de = parity.max_rel_deviation(pp[3].cpu().numpy(), ref[3].numpy(), w["variance_adaptor.energy_bins"].numpy(), valid)
    assert de <= parity.EDGE_REL, (what, "energy deviates from the oracle by more than the recorded noise bound", de)
    return {"edge_flips": fp[0] + fe[0], "max_rel_dev": max(dp, de), "energy": close(tf[3] * ec, ref[3].numpy(), MEL_TOL, what + " energy"),
            "mel": close(tf[0], ref[0].numpy(), MEL_TOL, what + " mel"),
            "postnet": close(tf[1], ref[1].numpy(), MEL_TOL, what + " postnet mel"), "out": out}


def test_full_size_vs_oracle_and_properties():
    """Config 2 at full size, every frame: HIP path vs the oracle run on this box's host cores, plus
    size-independent properties (mask/length consistency, zeroed pad frames of the mel projection input)."""
    from oracle import fs2_oracle as orc

    meta = dict(config="ljspeech", weight_seed=0, frames_per_phoneme=8.0, dur_weight_scale=0.25)
    cfg, sd, m = gpu_model(meta)
    w = orc.to_torch_weights(sd)
    inp, ref = _screened_full_batch(w, cfg, 16, 128, seed=3)
    r = _pinned_vs_oracle(m, w, cfg, inp, ref, "config 2 full size")
    out = r.pop("out")
    print("config 2 full size (B=16, L=128), every frame:", r)
    mel_lens = out[9].cpu().numpy()
    T = out[0].shape[1]
    assert T == mel_lens.max()
    assert np.array_equal(out[7].cpu().numpy(), np.arange(T)[None, :] >= mel_lens[:, None])
    assert np.array_equal(mel_lens, np.maximum(out[5].cpu().numpy().astype(np.int64), 0).sum(1))
    # p/e predictions are masked to exactly 0 on padded frames (model/modules.py:283-284)
    assert np.all(out[2].cpu().numpy()[out[7].cpu().numpy()] == 0.0)
    assert np.all(out[3].cpu().numpy()[out[7].cpu().numpy()] == 0.0)


@pytest.mark.parametrize("workload", ["cfg4_d512", "cfg5_longform"])
def test_full_size_cfg4_cfg5_vs_oracle(workload):
    """BASELINE configs 4 (d=512, 6+6 layers, 8 heads, B=64: 64 x ~1010 rows through the 512-wide tiles) and 5 (B=8,
    T~3900: 31 query tiles x 122 key tiles per head) at their FULL batch sizes, every frame, against the oracle with
    the bucket decisions pinned (the reference-generated pins cover B=8 / B=2 of these shapes)."""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import fs2_oracle as orc

    cfg_name, B, L, fpp = wl.WORKLOADS[workload]
    meta = dict(config=cfg_name, weight_seed=0, frames_per_phoneme=fpp, dur_weight_scale=0.25)
    cfg, sd, m = gpu_model(meta)
    w = orc.to_torch_weights(sd)
    inp, ref = _screened_full_batch(w, cfg, B, L, seed=0)
    r = _pinned_vs_oracle(m, w, cfg, inp, ref, workload)
    out = r.pop("out")
    print(workload, "B", B, "L", L, "T_pad", out[0].shape[1], "rows", B * out[0].shape[1], "valid frames", int(ref[9].sum()), r)
    assert out[0].shape[0] == B and (out[0].shape[1] > 3000 if workload == "cfg5_longform" else B * out[0].shape[1] > 60000)
    _MODEL.clear()  # 329 MB of d=512 weights + ~1 GB of scratch: give it back before the next test


def test_cfg5_longform_gaussian_full_size_vs_oracle():
    """BASELINE config 5 AS WORDED ("mel_len=4000, batch=8, Gaussian-upsample + variable-length masking stress"): the
    reference's GaussianUpsampling (model/modules.py:166-192) wired in as the length regulator (the §8 f1 extension), B=8,
    L=128, ~31 frames per phoneme -> T_pad ~3900, at FULL size, every frame of every output against the oracle with the
    bucket decisions pinned.  (w = [8, 128, ~3900] Gaussian weights per forward; 122 key tiles per attention sweep.)"""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import fs2_oracle as orc

    cfg_name, B, L, fpp = wl.WORKLOADS["cfg5_longform_gaussian"]
    assert cfg_name.endswith("+gaussian") and B == 8
    meta = dict(config="ljspeech", weight_seed=0, frames_per_phoneme=fpp, dur_weight_scale=0.25, length_regulator="gaussian")
    cfg, sd, m = gpu_model_for(meta)
    assert m._cfg.length_regulator == 1
    w = orc.to_torch_weights(sd)
    inp, ref = _screened_full_batch(w, cfg, B, L, seed=0, length_regulator="gaussian")
    r = _pinned_vs_oracle(m, w, cfg, inp, ref, "cfg5_longform_gaussian")
    out = r.pop("out")
    T = out[0].shape[1]
    print("cfg5_longform_gaussian B", B, "L", L, "T_pad", T, "mel_lens", out[9].cpu().tolist(), r)
    assert out[0].shape[0] == 8 and T > 3000 and len(set(out[9].cpu().tolist())) > 1  # long-form AND variable lengths
    # Gaussian regulator semantics at this size: frames past an utterance's own length are zeroed before the decoder,
    # so pitch / energy there are exactly 0 and the mask marks them
    pad = out[7].cpu().numpy()
    assert pad.any() and np.all(out[2].cpu().numpy()[pad] == 0.0) and np.all(out[3].cpu().numpy()[pad] == 0.0)
    # and it is NOT the hard regulator's output (the test would be vacuous if the switch were ignored)
    _, _, mh = gpu_model(dict(config="ljspeech", weight_seed=0, frames_per_phoneme=fpp, dur_weight_scale=0.25))
    with torch.no_grad():
        hard = mh(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3], p_targets=ref[2].cuda(), e_targets=ref[3].cuda())
    assert torch.equal(hard[9], out[9]) and float((hard[0] - out[0]).abs().max()) > 1e-2
    _MODEL.clear()


def test_cfg5_longform_ragged_vs_oracle():
    """Config 5's "variable-length masking stress" on the reference's real path (hard LengthRegulator): B=8, phoneme counts
    drawn in [L/8, L] with one utterance at L, 31 frames per phoneme -> T_pad ~3900 while the shortest utterances leave
    thousands of padded frames (dozens of fully padded GEMM tiles, attention query AND key tiles, masked LayerNorm rows).
    EVERY output row — padded ones included — against the oracle, bucket decisions pinned."""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import fs2_oracle as orc

    cfg_name, B, L, fpp = wl.WORKLOADS["cfg5_longform"]
    meta = dict(config=cfg_name, weight_seed=0, frames_per_phoneme=fpp, dur_weight_scale=0.25)
    cfg, sd, m = gpu_model(meta)
    w = orc.to_torch_weights(sd)

    def lens_fn(n):
        ln = np.random.RandomState(11).randint(L // 8, L + 1, size=n)
        ln[0::5] = L  # a few full-length candidates: whichever survives the screen sets T_pad ~ 3900
        return ln

    inp, ref = _screened_full_batch(w, cfg, B, L, seed=4, lens_fn=lens_fn)
    assert inp[2].max() == L, "screening dropped every full-length utterance"
    r = _pinned_vs_oracle(m, w, cfg, inp, ref, "cfg5_longform ragged")
    out = r.pop("out")
    lens = out[9].cpu().numpy()
    print("cfg5_longform ragged: src_lens", inp[2].tolist(), "mel_lens", lens.tolist(), "T_pad", out[0].shape[1], r)
    assert out[0].shape[1] > 3000 and lens.min() < out[0].shape[1] // 2  # long-form, and at least half of a row is padding
    _MODEL.clear()


def _plan_of(m, M, N, Cin, KW):
    """include/nar_fs2.h ns_plan_gemm as a dict, or None below the planner's range"""
    import ctypes as C

    o = (C.c_int32 * 8)()
    if not m._lib.ns_plan_gemm(int(M), int(N), int(Cin), int(KW), o):
        return None
    return {"main": (o[0], o[1], o[2]), "rem": (o[3], o[4], o[5]), "mfma_edge": o[6], "model_us": o[7]}


@pytest.mark.parametrize("B", [9, 11, 17, 20])
def test_planner_shapes_vs_oracle(B):
    """The launch plan's OWN shapes against the oracle: uniform batches of B x L=128 utterances at the LJSpeech config put
    B*T_pad between the steps of 256 workgroups, where the plan cuts GEMMs into a main + remainder launch / picks the 16-row
    tile family and cuts attention's key axis (include/nar_fs2.h ns_plan_gemm, ns_plan_attention_split; the reference's
    nn.Conv1d / bmm take any row count, transformer/SubLayers.py:87-95, transformer/Modules.py:14-25).  The test first asserts
    through the planning entry points that such a path IS taken at this B, then compares every frame, discrete decisions pinned."""
    from oracle import fs2_oracle as orc

    meta = dict(config="ljspeech", weight_seed=0, frames_per_phoneme=8.0, dur_weight_scale=0.25)
    cfg, sd, m = gpu_model(meta)
    w = orc.to_torch_weights(sd)
    inp, ref = _screened_full_batch(w, cfg, B, 128, seed=20 + B)
    T = int(ref[9].max())
    M = B * T
    t = cfg["transformer"]
    d, di, k1 = t["decoder_hidden"], t["conv_filter_size"], t["conv_kernel_size"][0]
    plans = {"w_1": _plan_of(m, M, di, d, k1), "postnet_mid": _plan_of(m, M, 512, 512, 5), "qkv": _plan_of(m, M, 3 * d, d, 1)}
    nsplit = int(m._lib.ns_plan_attention_split(B, T, t["decoder_head"], d // t["decoder_head"]))
    cut = [k for k, p in plans.items() if p and p["rem"][2] > 0]
    fine = [k for k, p in plans.items() if p and p["mfma_edge"] == 16]  # a tile of the 16-row family (height chosen for the row count)
    row_tile = int(m._lib.ns_plan_row_tile(M, d))  # the full-row (GEMM + LayerNorm epilogue) tile's height
    print(f"B={B} T_pad={T} rows={M} plans={plans} full-row tile {row_tile} attention key split={nsplit}")
    if row_tile != 32:
        fine.append("full_row")
    assert plans["w_1"] is not None, "B*T is inside the planner's range"
    assert cut or fine or nsplit > 1, (B, "neither a main + remainder cut, a 16-row tile nor a key split is planned at this shape", plans, nsplit)
    r = _pinned_vs_oracle(m, w, cfg, inp, ref, f"planner shape B={B}")
    out = r.pop("out")
    assert out[0].shape == (B, T, 80)
    print(f"B={B}:", r)


@pytest.mark.parametrize("cfg_name,B,height", [("tiny", 9, 48), ("tiny", 17, 80), ("tiny", 25, 112), ("tiny512", 9, 48), ("tiny512", 17, 32),
                                                ("tiny512", 25, 32)])
def test_full_row_tile_heights_vs_oracle(cfg_name, B, height):
    """Every height of the full-row (GEMM + LayerNorm / predictor-tail epilogue) tile the plan can pick — 48, 80 and 112 rows of the
    16-row family, one- and two-pass row epilogues — at both row widths (256 and 512 columns: the 1+1-layer fixtures models; at 512
    columns only 32 and 48 rows: the taller forms have no room for the chunked accumulation's second register set), each
    against the oracle on every frame with the bucket decisions pinned.  B x ~1010 rows put the fullest CU at 35.5 / 67 / 98.6 rows;
    ns_plan_row_tile must answer the height the case is named for (transformer/SubLayers.py:56-57,92-93; model/modules.py:245-286)."""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import fs2_oracle as orc

    meta = dict(config=cfg_name, weight_seed=0, frames_per_phoneme=8.0, dur_weight_scale=0.25)
    cfg, sd, m = gpu_model(meta)
    w = orc.to_torch_weights(sd)
    inp, ref = _screened_full_batch(w, cfg, B, 128, seed=40 + B)
    T = int(ref[9].max())
    d = cfg["transformer"]["decoder_hidden"]
    assert int(m._lib.ns_plan_row_tile(B * T, d)) == height, (B * T, d, int(m._lib.ns_plan_row_tile(B * T, d)))
    r = _pinned_vs_oracle(m, w, cfg, inp, ref, f"{cfg_name} B={B} full-row tile {height}")
    r.pop("out")
    print(f"{cfg_name} B={B} rows {B * T} full-row tile {height} x {d}:", r)
    _MODEL.clear()


class _ShardedGlobalPad:
    """Runs a batch as contiguous shards, one forward each on the one GPU, every shard padded to the GLOBAL longest mel
    (forward(max_mel_len=callable), the value sharding.global_max would all-reduce) and returns the concatenated 12-tuple:
    SURVEY.md section 8e's secondary parity statement, 'concatenated ranks must match the reference run on all N'."""

    def __init__(self, m, world):
        self.m, self.world = m, world

    def __call__(self, sp, tx, ln, L, p_targets=None, e_targets=None, **kw):
        from smart_nar_fast_tts_amd import sharding

        n = int(tx.shape[0])
        bounds = [sharding.shard_bounds(n, self.world, r) for r in range(self.world)]
        cut = lambda t, lo, hi: None if t is None else t[lo:hi]  # noqa: E731
        # pass 1: every rank's local longest (what each would contribute to the all-reduce MAX)
        local = [int(self.m(sp[lo:hi], tx[lo:hi], ln[lo:hi], L, **kw)[9].max()) for lo, hi in bounds]
        G = max(local)
        self.local_max, self.global_max = local, G
        outs = [self.m(sp[lo:hi], tx[lo:hi], ln[lo:hi], L, max_mel_len=lambda t, G=G: max(int(t), G),
                       p_targets=cut(p_targets, lo, hi), e_targets=cut(e_targets, lo, hi), **kw) for lo, hi in bounds]
        return tuple(torch.cat([o[i] for o in outs]) if torch.is_tensor(outs[0][i]) else None for i in range(12))


@pytest.mark.parametrize("rows", ["packed", "dense"])
def test_global_pad_shards_match_the_oracle_on_the_full_batch(rows):
    """SURVEY.md section 8e, secondary parity: ONE ragged batch of 12, the oracle on the FULL batch; the HIP path on two
    contiguous shards of 6 with max_mel_len = the global longest mel (sharding.global_max's value); the concatenation
    against the oracle: durations / frame counts / masks identical, mel < 1e-3 with the bucket decisions pinned, on EVERY
    utterance - including shard 1's longest, which is un-padded per shard but padded in the full batch (the utterance the
    reference itself computes differently in the two settings: model/modules.py:128-137,201-230, utils/tools.py:288-306)."""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import fs2_oracle as orc

    meta = dict(config="ljspeech", weight_seed=0, frames_per_phoneme=8.0, dur_weight_scale=0.25)
    cfg, sd, m = gpu_model(meta)
    w = orc.to_torch_weights(sd)
    lens = np.array([100, 37, 64, 12, 81, 55, 70, 9, 33, 48, 70, 21])  # shard 0 holds the global longest, shard 1's longest is 70
    seed, inp, ref = _oracle_case_with_margin(w, cfg, len(lens), 100, seed=77, lens=lens)
    keep = m.packed_rows
    m.packed_rows = rows == "packed"
    try:
        two = _ShardedGlobalPad(m, 2)
        r = _pinned_vs_oracle(two, w, cfg, inp, ref, f"global-pad shards ({rows})")
        out = r.pop("out")
    finally:
        m.packed_rows = keep
    T = int(ref[9].max())
    assert two.global_max == T and min(two.local_max) < T, (two.local_max, T)  # shard 1 really was padded beyond its own longest
    assert out[0].shape == tuple(ref[0].shape) and out[7].shape == tuple(ref[7].shape)
    print(f"global-pad, 2 shards of 6 ({rows} rows): local max {two.local_max} -> global {T};", r)
    # and the per-shard mode is NOT this (the statement is not vacuous): shard 1 alone pads to its own longest only
    with torch.no_grad():
        alone = m(dev(inp[0][6:]), dev(inp[1][6:]), dev(inp[2][6:]), inp[3])
    assert alone[0].shape[1] == two.local_max[1] < T


def test_edge_cases():
    """Ragged / minimal inputs the reference handles: B=1,L=1; heavy phoneme-side padding; p/e control.  Every case is
    compared (inputs are re-drawn until no oracle duration sits on a rounding boundary) and the count is asserted."""
    from oracle import fs2_oracle as orc

    meta = dict(config="tiny", weight_seed=0, frames_per_phoneme=4.0, dur_weight_scale=0.25)
    cfg, sd, m = gpu_model(meta)
    w = orc.to_torch_weights(sd)
    cases = [(1, 1, None), (2, 5, [5, 1]), (4, 33, [33, 2, 17, 32]), (3, 130, [130, 64, 129])]
    ran = 0
    for B, L, lens in cases:
        for pc, ec in ((1.0, 1.0), (1.3, 0.7)):
            seed, inp, ref = _oracle_case_with_margin(w, cfg, B, L, seed=21, lens=lens, p_control=pc, e_control=ec)
            r = _pinned_vs_oracle(m, w, cfg, inp, ref, f"B={B} L={L} p_control={pc} e_control={ec}", p_control=pc, e_control=ec)
            r.pop("out")
            ran += 1
            print(f"edge case B={B} L={L} lens={lens} controls=({pc},{ec}) seed={seed}:", r)
    assert ran == 2 * len(cases), ran


def test_all_zero_durations_give_empty_output():
    """All durations 0 -> T = 0 (SURVEY.md §8b Errors): shapes [B,0,80], no launch, no crash."""
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg = wl.model_config("tiny")
    sd = wl.synth_state_dict(cfg, seed=0, frames_per_phoneme=0.0)  # bias log(1) = 0 -> round(exp(~0)-1) = 0
    sd["variance_adaptor.duration_predictor.linear_layer.weight"] *= 0.0
    m = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda")
    m.load_state_dict(sd)
    inp = wl.synth_inputs(2, 6, seed=1)
    out = m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3])
    assert tuple(out[0].shape) == (2, 0, 80) and tuple(out[1].shape) == (2, 0, 80)
    assert out[9].cpu().tolist() == [0, 0]


def test_errors_are_loud():
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg = wl.model_config("tiny")
    m = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda")
    inp = wl.synth_inputs(1, 4, seed=1)
    with pytest.raises(RuntimeError, match="weights not loaded"):
        m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3])
    sd = wl.synth_state_dict(cfg)
    bad = dict(sd)
    bad["mel_linear.weight"] = bad["mel_linear.weight"][:, :100]
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(bad)
    bad = dict(sd)
    bad["not.a.key"] = np.zeros(3, np.float32)
    with pytest.raises(RuntimeError, match="unexpected key"):
        m.load_state_dict(bad)
    with pytest.raises(RuntimeError, match="cuda"):
        FastSpeech2Align(wl.preprocess_config(), cfg).to("cpu")


def gpu_model_for(meta):
    """Like gpu_model, for fixtures that also choose feature levels / the length regulator."""
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg, sd = weights_for(meta)
    cfg = dict(cfg, length_regulator=meta.get("length_regulator", "hard"))
    pc = wl.preprocess_config(meta.get("pitch_level", "frame_level"), meta.get("energy_level", "frame_level"))
    m = FastSpeech2Align(pc, cfg).to("cuda").eval()
    m.load_state_dict(sd)
    return cfg, sd, m


@pytest.mark.parametrize("name", ["e2e_tiny_phoneme_level", "e2e_tiny_pitch_phoneme_energy_frame",
                                  "e2e_tiny_pitch_frame_energy_phoneme", "e2e_tiny_gaussian_wired"])
def test_feature_levels_and_gaussian_regulator(name):
    """§8 f4: phoneme_level pitch/energy (predicted on the encoder output, [B,L] predictions) in all three mixes, with
    p/e_control != 1; §8 f1: the reference's GaussianUpsampling module wired in as the length regulator (extension)."""
    meta, z = load_golden(name)
    cfg, sd, m = gpu_model_for(meta)
    with torch.no_grad():
        out = m(dev(z["speakers"]), dev(z["texts"]), dev(z["in_src_lens"]), int(meta["L"]),
                p_control=meta.get("p_control", 1.0), e_control=meta.get("e_control", 1.0))
    torch.cuda.synchronize()
    print(name, check_12tuple(out, z))


def test_encoder_longer_than_max_seq_len():
    """L > max_seq_len: TxtEncoder rebuilds the position table too (transformer/Models.py:82-87); also exercises
    long single-utterance decoding (T ~ 4000 > max_seq_len) against the oracle."""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import fs2_oracle as orc

    meta = dict(config="tiny", weight_seed=0, frames_per_phoneme=4.0, dur_weight_scale=0.25)
    cfg, sd, m = gpu_model(meta)
    inp = wl.synth_inputs(1, 1030, seed=5)
    with torch.no_grad():
        ref = orc.forward(orc.to_torch_weights(sd), cfg, torch.from_numpy(inp[0]), torch.from_numpy(inp[1]),
                          torch.from_numpy(inp[2]), inp[3])
        out = m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3])
    close(out[4], ref[4].numpy(), 1e-4, "log durations (L=1030)")
    half = np.abs((np.exp(ref[4].numpy().astype(np.float64)) - 1.0) % 1.0 - 0.5)
    flips = out[5].cpu().numpy() != ref[5].numpy()
    assert np.all(half[flips] < 5e-5)
    if not flips.any():
        with torch.no_grad():
            tf = m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3], p_targets=ref[2].cuda(), e_targets=ref[3].cuda())
        close(tf[1], ref[1].numpy(), MEL_TOL, "postnet mel (L=1030, T~4000)")


def test_max_mel_len_global_pad_mode():
    """Extension used by multi-GPU global-pad mode: max_mel_len > max(mel_lens) pads and masks the mel axis.
    Utterances that were already padded keep bit-identical results; shapes follow max_mel_len."""
    meta, z = load_golden("e2e_tiny_padded_src")
    cfg, sd, m = gpu_model(meta)
    m.packed_rows = False  # the bit-identity statements below are about the padded GRID (capacity mode always runs on it)
    try:
        base = run_gpu(m, z, meta)
        T = base[0].shape[1]
        with torch.no_grad():
            padded = m(dev(z["speakers"]), dev(z["texts"]), dev(z["in_src_lens"]), int(meta["L"]), max_mel_len=T + 9)
            via_fn = m(dev(z["speakers"]), dev(z["texts"]), dev(z["in_src_lens"]), int(meta["L"]), max_mel_len=lambda t: int(t) + 9)
            m.packed_rows = True  # the synchronous path's default: packed rows, same values up to fp32 summation order
            via_packed = m(dev(z["speakers"]), dev(z["texts"]), dev(z["in_src_lens"]), int(meta["L"]), max_mel_len=lambda t: int(t) + 9)
    finally:
        m.packed_rows = True
    assert via_packed[1].shape == padded[1].shape and torch.equal(via_packed[7], padded[7]) and torch.equal(via_packed[9], padded[9])
    close(via_packed[1], padded[1].cpu().numpy(), 2e-5, "global-pad mode on packed rows vs the grid (postnet mel, every frame)")
    close(via_packed[0], padded[0].cpu().numpy(), 2e-5, "global-pad mode on packed rows vs the grid (mel, every frame)")
    assert padded[0].shape[1] == T + 9 and padded[7].shape[1] == T + 9
    assert torch.equal(padded[1], via_fn[1])
    assert torch.equal(padded[9], base[9])
    lens = base[9].cpu().numpy()
    assert np.array_equal(padded[7].cpu().numpy(), np.arange(T + 9)[None, :] >= lens[:, None])
    for b in range(len(lens)):
        if lens[b] < T:  # had padding before: unchanged
            assert torch.equal(padded[1][b, :lens[b]], base[1][b, :lens[b]])
    with pytest.raises(ValueError, match="smaller than the longest"):
        m(dev(z["speakers"]), dev(z["texts"]), dev(z["in_src_lens"]), int(meta["L"]), max_mel_len=lambda t: T - 1)


def test_capacity_mode_is_sync_free_and_bit_identical():
    """max_mel_len=<int> with the explicit opt-in async_status=True (model/modules.py:128-131,204-213 `max_len` semantics):
    phase 2 is enqueued right behind phase 1, nothing on the host waits for mel_lens.  With the capacity equal to the longest
    utterance every output is BIT-identical to the synchronous path; what the synchronous path raises on the spot arrives
    through the returned output's own status (out.check()), per call."""
    from unittest import mock

    from smart_nar_fast_tts_amd import _lib
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    meta, z = load_golden("e2e_tiny_padded_src")
    cfg, sd, m = gpu_model(meta)
    m.packed_rows = False  # capacity mode runs on the padded grid (the host never learns the lengths): compare grid with grid
    base = run_gpu(m, z, meta)
    T = base[0].shape[1]
    nb = len(z["in_src_lens"])
    args = (dev(z["speakers"]), dev(z["texts"]), dev(z["in_src_lens"]), int(meta["L"]))
    with torch.no_grad(), mock.patch.object(FastSpeech2Align, "_wait_phase1", side_effect=AssertionError("host waited")), \
            mock.patch.object(torch.cuda.Event, "synchronize", side_effect=AssertionError("host waited")), \
            mock.patch.object(torch.cuda, "synchronize", side_effect=AssertionError("host waited")):
        cap = m(*args, max_mel_len=T, async_status=True)   # would raise if the forward waited for phase 1 in any way
    assert cap.check() == [0] * nb and m.check_status(cap) == [0] * nb and m.check_status() == [0] * nb
    assert len(cap) == 12 and isinstance(cap, tuple)
    for i in (0, 1, 2, 3, 4, 5, 6, 7, 9):
        assert torch.equal(cap[i], base[i]), NAMES[i]
    # a larger capacity: same as the synchronous padded run — through a callable and through a plain int
    with torch.no_grad():
        cap9 = m(*args, max_mel_len=T + 9, async_status=True)
        ref9 = m(*args, max_mel_len=lambda t: int(t) + 9)
        int9 = m(*args, max_mel_len=T + 9)  # an int WITHOUT the opt-in is synchronous
    assert all(torch.equal(cap9[i], ref9[i]) and torch.equal(int9[i], ref9[i]) for i in (0, 1, 2, 3, 7, 9)) and cap9.check() == [0] * nb
    # too small a capacity: every utterance longer than it is reported, the others are bit-identical to a run at that padding
    lens = base[9].cpu().numpy()
    Tcut = int(np.sort(lens)[-2]) if len(lens) > 1 and np.sort(lens)[-2] < T else T - 1
    with torch.no_grad():
        cut = m(*args, max_mel_len=Tcut, async_status=True)
        ok_again = m(*args, max_mel_len=T, async_status=True)  # a later forward must not overwrite the earlier call's status
    words = cut.status.cpu().numpy()
    assert np.array_equal(words & _lib.STATUS_TRUNCATED, (lens > Tcut).astype(np.int32)) and cut[0].shape[1] == Tcut
    assert torch.isfinite(cut[1]).all() and torch.equal(cut[9], base[9])
    with pytest.raises(ValueError, match="cut off"):
        cut.check()
    assert ok_again.check() == [0] * nb
    # ... while the same int WITHOUT the opt-in raises on the spot, like every synchronous forward
    with pytest.raises(ValueError, match="smaller than the longest"):
        m(*args, max_mel_len=Tcut)
    with pytest.raises(ValueError, match="async_status"):
        m(*args, async_status=True)
    # a bad token id: the synchronous paths raise IndexError on the spot, capacity mode through the output's status
    bad = z["texts"].copy()
    bad[0, 0] = 100000
    with pytest.raises(IndexError):
        m(args[0], dev(bad), args[2], args[3])
    with pytest.raises(IndexError):
        m(args[0], dev(bad), args[2], args[3], max_mel_len=T)
    with torch.no_grad():
        ob = m(args[0], dev(bad), args[2], args[3], max_mel_len=T, async_status=True)
    assert ob.status.cpu().numpy()[0] & _lib.STATUS_BAD_TOKEN
    with pytest.raises(IndexError):
        ob.check()
    # the C-ABI refuses to run without a status buffer (a C caller cannot truncate silently)
    rc = m._lib.ns_forward_mel(m._h, 1, 1, 1, None, 1.0, 1.0, None, None, None, None, 0, None, None, None, None, None, None, None)
    assert rc != 0 and b"status" in m._lib.ns_last_error()
    m.packed_rows = True


def test_fresh_workspace_after_a_length_hint_is_large_enough():
    """ns_decoder_ws_bytes is not monotonic in T (a shorter mel axis can take attention's split-key path, whose partials
    outweigh the rest), and the synchronous forward sizes phase 2's scratch BEFORE it knows T, from the previous forward of
    the same shape plus slack.  With a fresh workspace (new stream, release_workspaces(), LRU eviction) that guess alone was
    too small: B=32 at T ~ 120 needed 119 MB against 81 MB allocated and forward() raised 'workspace too small'."""
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg = wl.model_config("ljspeech")
    m = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda:0").eval()
    m.load_state_dict(wl.synth_state_dict(cfg, seed=0, frames_per_phoneme=7.0))
    hit = 0
    for B, L in ((32, 15), (8, 120), (16, 48)):
        sp, tx, ln, Lm = wl.synth_inputs(B, L, seed=3)
        a = [dev(x) for x in (sp, tx, ln)]
        with torch.no_grad():
            first = m(a[0], a[1], a[2], Lm)
            T = int(first[0].shape[1])
            Tc = T + max(8, T >> 3)
            need_T, need_Tc = m._ws_bytes("dec", B, L, T), m._ws_bytes("dec", B, L, Tc)
            hit += int(need_T > int(need_Tc * 1.25) + 256)  # the case that used to raise
            m.release_workspaces()
            again = m(a[0], a[1], a[2], Lm)          # hint path, fresh scratch
            s2 = torch.cuda.Stream()
            with torch.cuda.stream(s2):               # and on a stream that has never been used
                third = m(a[0], a[1], a[2], Lm)
            torch.cuda.synchronize()
        assert torch.equal(first[1], again[1]) and torch.equal(first[1], third[1]) and torch.equal(first[9], again[9])
    print(f"shapes whose exact-T workspace exceeds the hinted allocation: {hit} of 3")


def test_rejected_state_dict_leaves_the_loaded_model_usable():
    """load good weights, attempt a bad load (unexpected key / wrong shape), then run — the forward
    still works and its output is bit-identical to before."""
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    meta, z = load_golden("e2e_tiny_padded_src")
    cfg, sd = weights_for(meta)
    m = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
    m.load_state_dict(sd)
    before = run_gpu(m, z, meta)
    for bad, msg in (({"not.a.key": np.zeros(3, np.float32)}, "unexpected key"),
                     ({"mel_linear.bias": np.zeros(81, np.float32)}, "size mismatch")):
        with pytest.raises(RuntimeError, match=msg):
            m.load_state_dict(dict({"mel_linear.weight": np.ones((80, 256), np.float32)}, **bad))
    after = run_gpu(m, z, meta)
    assert torch.equal(before[1], after[1]) and torch.equal(before[9], after[9])
    # and a partial (valid) update does take effect
    m.load_state_dict({"mel_linear.bias": np.asarray(sd["mel_linear.bias"]) + 1.0})
    moved = run_gpu(m, z, meta)
    assert torch.allclose(moved[0], before[0] + 1.0, atol=1e-5)


def test_random_shapes_vs_oracle():
    """Fuzz over batch shapes (tile tails of every GEMM variant, ragged lengths, tiny and >128-row utterances,
    1..3 frames per phoneme): HIP path vs the oracle on the same seeded inputs, discrete decisions pinned with targets."""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import fs2_oracle as orc
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    rs = np.random.RandomState(2024)
    cfg = wl.model_config("tiny")
    checked, worst = 0, 0.0
    for fpp in (1.0, 3.0):
        sd = wl.synth_state_dict(cfg, seed=1, frames_per_phoneme=fpp)
        w = orc.to_torch_weights(sd)
        m = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
        m.load_state_dict(sd)
        for _ in range(6):
            B = int(rs.randint(1, 7))
            L = int(rs.choice([1, 2, 5, 31, 32, 33, 63, 64, 65, 100, 129, 200]))
            lens = np.maximum(1, rs.randint(1, L + 1, size=B))
            lens[rs.randint(B)] = L
            # re-draw the token ids (not the shape) until no oracle duration sits on a rounding boundary: every shape is compared
            seed, inp, ref = _oracle_case_with_margin(w, cfg, B, L, seed=int(rs.randint(1 << 20)), lens=lens)
            if int(ref[9].max()) == 0:  # all durations zero: T = 0, nothing to compare beyond the shapes
                with torch.no_grad():
                    out = m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3])
                assert out[0].shape[1] == 0 and np.array_equal(out[5].cpu().numpy(), ref[5].numpy())
            else:
                r = _pinned_vs_oracle(m, w, cfg, inp, ref, f"fpp={fpp} B={B} L={L} lens={lens.tolist()}")
                worst = max(worst, r["mel"], r["postnet"])
            checked += 1
    print("random shapes compared:", checked, "worst mel/postnet max-abs", worst)
    assert checked == 12, checked


def test_large_batch_replicas_are_identical():
    """BASELINE config 3's global batch (and beyond) on ONE GPU: 192 utterances = the seeded 16-utterance batch
    tiled 12x, so B*T_pad is ~194k rows and every index computation in the kernels runs far from the sizes the other
    tests use.  Utterances interact only through padding, so (1) every replica must come out bit-identical to the
    first one (same kernels, same summation order, different rows) and (2) the replicas must agree with the plain
    16-utterance run to fp32 noise (that run takes different tile shapes, so the order of the K sums differs) with
    identical integer durations."""
    import smart_nar_fast_tts_amd.workload as wl

    meta = dict(config="ljspeech", weight_seed=0, frames_per_phoneme=8.0, dur_weight_scale=0.25)
    cfg, sd, m = gpu_model(meta)
    sp, tx, ln, L = wl.synth_inputs(16, 128, seed=5)
    R = 12
    with torch.no_grad():
        small = m(dev(sp), dev(tx), dev(ln), L)
        # the oracle-free way to pin the discrete bucket choices on both runs: hand the small run's pitch/energy in
        big = m(dev(np.tile(sp, R)), dev(np.tile(tx, (R, 1))), dev(np.tile(ln, R)), L,
                p_targets=small[2].repeat(R, 1), e_targets=small[3].repeat(R, 1))
        small_p = m(dev(sp), dev(tx), dev(ln), L, p_targets=small[2], e_targets=small[3])
    torch.cuda.synchronize()
    assert big[0].shape[0] == 16 * R and big[0].shape[1] == small[0].shape[1]
    assert torch.equal(big[5][:16], small[5]) and torch.equal(big[9][:16], small[9])
    for i in (0, 1, 2, 3, 4, 5, 6, 7, 9):
        first = big[i][:16]
        for r in range(1, R):
            assert torch.equal(big[i][16 * r:16 * (r + 1)], first), (NAMES[i], "replica", r, "differs from replica 0")
    valid = ~small[7].cpu().numpy()
    for i in (0, 1):
        err = (big[i][:16] - small_p[i]).abs().cpu().numpy()[valid].max()
        assert err < 2e-5, (NAMES[i], err)
    # 17 and 21 copies of ONE utterance: B*T_pad is a little more than one full round of the 256x256 tile, so the k=9 GEMM is
    # cut into that round + a remainder launch (gemm_conv.hip split plan) — the copies behind the cut must carry the same bits
    T1 = int(small[9][0])  # the copies' own length is their T_pad
    for R1 in (17, 21):
        with torch.no_grad():
            one = m(dev(np.tile(sp[:1], R1)), dev(np.tile(tx[:1], (R1, 1))), dev(np.tile(ln[:1], R1)), L,
                    p_targets=small[2][:1, :T1].repeat(R1, 1), e_targets=small[3][:1, :T1].repeat(R1, 1))
        assert R1 * T1 > 16384 and one[0].shape[1] == T1
        torch.cuda.synchronize()
        for i in (0, 1, 2, 3, 4, 5, 9):
            assert all(torch.equal(one[i][r], one[i][0]) for r in range(1, R1)), (NAMES[i], R1, "copies differ across the split")


def test_ragged_batch_vs_oracle():
    """Ragged batch at the LJSpeech widths: one long utterance sets T_pad ~ 1000 while the others leave hundreds of
    padded frames (whole GEMM tiles and attention query/key tiles of padding, masked LayerNorm rows that are written
    without being read).  Every output row — padded ones included — must match the oracle."""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import fs2_oracle as orc

    meta = dict(config="ljspeech", weight_seed=0, frames_per_phoneme=8.0, dur_weight_scale=0.25)
    cfg, sd, m = gpu_model(meta)
    lens = np.array([128, 10, 64, 1, 100, 33, 127, 17])
    inp = wl.synth_inputs(len(lens), 128, seed=9, src_lens=lens)
    w = orc.to_torch_weights(sd)
    with torch.no_grad():
        ref = orc.forward(w, cfg, torch.from_numpy(inp[0]), torch.from_numpy(inp[1]), torch.from_numpy(inp[2]), inp[3])
        out = m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3])
    assert np.array_equal(out[5].cpu().numpy(), ref[5].numpy()), "durations differ"
    assert np.array_equal(out[9].cpu().numpy(), ref[9].numpy())
    assert int(ref[9].max()) > 900 and int(ref[9].min()) < 16
    with torch.no_grad():
        tf = m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3], p_targets=ref[2].cuda(), e_targets=ref[3].cuda())
    close(tf[4], ref[4].numpy(), 1e-4, "log durations")
    close(tf[3], ref[3].numpy(), MEL_TOL, "energy")
    print("ragged: mel", close(tf[0], ref[0].numpy(), MEL_TOL, "mel (all rows, padded included)"),
          "postnet", close(tf[1], ref[1].numpy(), MEL_TOL, "postnet mel (all rows, padded included)"))


def test_packed_rows_match_the_dense_grid_on_ragged_batches():
    """include/nar_fs2.h ns_forward_mel_packed: phase 2 of a variable-length batch runs on the utterances' windows
    (min(len + 20, T) frames each) laid end to end instead of the reference's padded [B, T] grid
    (transformer/Layers.py:43,46, model/modules.py:283-284: padded frames are computed, then zeroed or ignored).
    Against the dense grid of the same build, bucket decisions pinned by the dense run's own pitch / energy values: every
    returned tensor on EVERY frame, padded ones included — integers and masks exact, floats within fp32 summation noise
    (a smaller launch may pick another tile shape) — and the packed run must really have run on fewer rows."""
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cases = [("ljspeech", 8.0, np.array([128, 10, 64, 1, 100, 33, 127, 17, 128, 5, 77, 2]), 128, {}),
             ("ljspeech", 31.0, np.array([128, 16, 90, 40, 128, 61, 20, 110]), 128, {}),   # long-form: T_pad ~ 3900
             ("ljspeech", 3.0, np.array([60, 5, 20, 2, 40]), 60, {}),                       # windows of a few dozen frames
             ("d512", 8.0, np.array([100, 30, 128, 64, 12, 90]), 128, {}),
             # BASELINE config 5 as worded: Gaussian upsampling AND variable lengths (the regulator writes into the windows)
             ("ljspeech", 31.0, np.array([128, 16, 90, 40, 128, 61, 20, 110]), 128, {"length_regulator": "gaussian"}),
             ("ljspeech", 8.0, np.array([100, 7, 64, 1, 33]), 100, {"length_regulator": "gaussian"})]
    for cfg_name, fpp, lens, L, extra in cases:
        _MODEL.clear()
        meta = dict(config=cfg_name, weight_seed=0, frames_per_phoneme=fpp, dur_weight_scale=0.25)
        cfg, sd = weights_for(meta)
        cfg = dict(cfg, **extra)
        inp = wl.synth_inputs(len(lens), L, seed=9, src_lens=lens)
        args = (dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3])
        models = {}
        for mode in ("dense", "packed"):
            m = FastSpeech2Align(wl.preprocess_config(), dict(cfg, padded_rows=mode)).to("cuda").eval()
            m.load_state_dict(sd)
            models[mode] = m
        with torch.no_grad():
            free = models["dense"](*args)
            pin = dict(p_targets=free[2].clone(), e_targets=free[3].clone())
            dense = models["dense"](*args, **pin)
            rows_dense = models["dense"]._lib.ns_last_phase2_rows(models["dense"]._h)
            packed = models["packed"](*args, **pin)
            rows_packed = models["packed"]._lib.ns_last_phase2_rows(models["packed"]._h)
            packed_free = models["packed"](*args)
            torch.cuda.synchronize()
        B, T = dense[0].shape[0], dense[0].shape[1]
        ml = dense[9].cpu().numpy()
        assert rows_dense == B * T and rows_packed == int(np.minimum(ml + 20, T).sum()) and rows_packed < 0.9 * rows_dense, \
            (cfg_name, rows_dense, rows_packed)
        for i in (4, 5, 6, 7, 9):
            assert torch.equal(packed[i], dense[i]) and torch.equal(packed_free[i], free[i]), (cfg_name, NAMES[i])
        worst = {}
        for i in (0, 1, 2, 3):
            assert packed[i].shape == dense[i].shape
            worst[NAMES[i]] = float((packed[i] - dense[i]).abs().max())
            assert torch.isfinite(packed[i]).all()
        print("packed vs dense", cfg_name, extra, "fpp", fpp, "rows", rows_packed, "of", rows_dense, worst)
        # (the packed path must not hide behind the bucket-edge bound — same arithmetic per row, so it agrees with the
        #  grid to fp32 summation order: measured worst 2.4e-6 on mel / PostNet mel over these six batches, bound 5e-6)
        assert worst["output"] < 5e-6 and worst["postnet_output"] < 5e-6 and worst["e_predictions"] < 2e-4, worst
        # with targets the prediction itself is returned: pitch does not depend on the targets, energy on the pitch bucket only
        assert worst["p_predictions"] < 2e-3 * max(1.0, float(dense[2].abs().max())), worst
        # free-running: the packed run takes its own bucket decisions; frame counts and masks still agree, values stay finite
        assert torch.isfinite(packed_free[1]).all()
    _MODEL.clear()


def test_bf16x3_mode_vs_oracle_and_fp32_path():
    """OPT-IN precision mode (ns_config.matmul_bf16x3, csrc/gemm_bf16x3.hip): the decoder FFN k=9 convolutions and the
    PostNet 512->512 convolutions computed from an exact 3-way bf16 split on the bf16 matrix cores.  Config 2 at full size:
    every discrete output identical to the oracle's (they are all upstream of those layers), mel / PostNet mel within the
    same 1e-3 bar with the buckets pinned, and within fp32 noise of the exact-fp32 HIP path — but NOT bit-identical to it
    (if it were, the mode would not have run)."""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import fs2_oracle as orc
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    meta = dict(config="ljspeech", weight_seed=0, frames_per_phoneme=8.0, dur_weight_scale=0.25)
    cfg, sd, m32 = gpu_model(meta)
    w = orc.to_torch_weights(sd)
    inp, ref = _screened_full_batch(w, cfg, 16, 128, seed=3)
    mb3 = FastSpeech2Align(wl.preprocess_config(), dict(cfg, matmul="bf16x3")).to("cuda").eval()
    mb3.load_state_dict(sd)
    assert mb3._lib.ns_arena_bytes(mb3._h) > m32._lib.ns_arena_bytes(m32._h)  # the weight planes are there
    r = _pinned_vs_oracle(mb3, w, cfg, inp, ref, "bf16x3 mode, config 2 full size")
    r.pop("out")
    with torch.no_grad():
        a = m32(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3], p_targets=ref[2].cuda(), e_targets=ref[3].cuda())
        b = mb3(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3], p_targets=ref[2].cuda(), e_targets=ref[3].cuda())
    for i in (2, 3, 4, 5, 9):  # pitch, energy, log durations, durations, frame counts: untouched by the mode
        assert torch.equal(a[i], b[i]), NAMES[i]
    d = float((a[1] - b[1]).abs().max())
    assert 0.0 < d < 5e-5, d
    print("bf16x3 mode vs oracle:", r, " vs the exact-fp32 HIP path, PostNet mel max-abs:", d)
    # small launches fall back to the fp32 kernels: a tiny batch through the bf16x3 model is bit-identical to the fp32 model
    sp, tx, ln, L = wl.synth_inputs(2, 20, seed=1)
    with torch.no_grad():
        assert torch.equal(m32(dev(sp), dev(tx), dev(ln), L)[1], mb3(dev(sp), dev(tx), dev(ln), L)[1])


def test_phase1_on_packed_phoneme_rows():
    """include/nar_fs2.h ns_forward_durations_packed: with src_lens on the HOST (a CPU tensor: what a caller that collates on the
    host holds, dataset.py:182-191) ragged batches run the encoder and the duration predictor on packed phoneme rows —
    min(src_len + 2, L) rows per utterance instead of the L the reference computes and then zeroes
    (transformer/Models.py:73-100, transformer/Layers.py:43,46).  Checked: (1) the reference's own fixture with phoneme-side
    padding, bit for bit on durations and frame counts; (2) ragged batches against the grid run of the same
    build: integers and masks exact, log-durations / mel within fp32 summation noise, the row count really smaller;
    (3) no packing where it is not exact or does not pay (phoneme_level features, uniform lengths, device src_lens)."""
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    # (1) F3(a): the reference's padded-source fixture ("always": the automatic rule only packs when a step of 256 workgroups is saved)
    meta, z = load_golden("e2e_tiny_padded_src")
    cfg, sd = weights_for(meta)
    m = FastSpeech2Align(wl.preprocess_config(), dict(cfg, phase1_packing="always")).to("cuda").eval()
    m.load_state_dict(sd)
    L = int(meta["L"])
    lens = z["in_src_lens"]
    with torch.no_grad():
        out = m(dev(z["speakers"]), dev(z["texts"]), torch.from_numpy(np.ascontiguousarray(lens)), L)
    rows1 = int(m._lib.ns_last_phase1_rows(m._h))
    expect_rows = int(np.minimum(lens + 2, L).sum())
    assert (rows1 == expect_rows) == (expect_rows * 10 <= len(lens) * L * 9), (rows1, expect_rows, len(lens) * L)
    assert np.array_equal(out[5].cpu().numpy(), z["d_rounded"]) and np.array_equal(out[9].cpu().numpy(), z["mel_lens"])
    assert np.array_equal(out[6].cpu().numpy(), z["src_masks"])
    close(out[4], z["log_d_predictions"], 2e-5, "log_d on packed phoneme rows vs the reference")
    assert out[8] is not None and np.array_equal(np.asarray(out[8]), lens)  # src_lens is passed through as it came

    # (2) ragged batches, packed phase 1 (host lens) against the grid (device lens) of the same model
    for cfg_name, fpp, lens, L, extra in (("ljspeech", 8.0, np.array([128, 10, 64, 1, 100, 33, 127, 17, 128, 5, 77, 2, 60, 90, 31, 111]), 128, {}),
                                          ("ljspeech", 3.0, np.array([60, 5, 20, 2, 40]), 60, {}),
                                          ("d512", 8.0, np.array([100, 30, 128, 64, 12, 90]), 128, {}),
                                          ("ljspeech", 8.0, np.array([100, 7, 64, 1, 33]), 100, {"length_regulator": "gaussian"})):
        _MODEL.clear()
        cfg, sd = weights_for(dict(config=cfg_name, weight_seed=0, frames_per_phoneme=fpp, dur_weight_scale=0.25))
        m = FastSpeech2Align(wl.preprocess_config(), dict(cfg, phase1_packing="always", **extra)).to("cuda").eval()
        m.load_state_dict(sd)
        inp = wl.synth_inputs(len(lens), L, seed=9, src_lens=lens)
        host_lens = torch.from_numpy(np.ascontiguousarray(inp[2]))
        with torch.no_grad():
            grid = m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3])
            rows_grid = int(m._lib.ns_last_phase1_rows(m._h))
            pin = dict(p_targets=grid[2].clone(), e_targets=grid[3].clone())
            grid_p = m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3], **pin)
            pk = m(dev(inp[0]), dev(inp[1]), host_lens, inp[3], **pin)
            rows_pk = int(m._lib.ns_last_phase1_rows(m._h))
            pk_list = m(dev(inp[0]), dev(inp[1]), [int(v) for v in inp[2]], inp[3], **pin)  # a plain list works too
        assert rows_grid == len(lens) * L and rows_pk == int(np.minimum(lens + 2, L).sum()) and rows_pk < 0.9 * rows_grid, (cfg_name, rows_grid, rows_pk)
        for i in (5, 6, 7, 9):
            assert torch.equal(pk[i], grid_p[i]) and torch.equal(pk_list[i], grid_p[i]), (cfg_name, NAMES[i])
        close(pk[4], grid_p[4].cpu().numpy(), 2e-5, f"{cfg_name}: log_d, packed phoneme rows vs grid")
        close(pk[0], grid_p[0].cpu().numpy(), 5e-5, f"{cfg_name}: mel, packed phoneme rows vs grid")
        close(pk[1], grid_p[1].cpu().numpy(), 5e-5, f"{cfg_name}: postnet mel, packed phoneme rows vs grid")
        # padded phonemes: log_d exactly 0 (masked_fill), like the grid
        pad = grid_p[6].cpu().numpy()
        assert np.all(pk[4].cpu().numpy()[pad] == 0.0)

    # (3) where it must not pack
    _MODEL.clear()
    cfg, sd = weights_for(dict(config="tiny", weight_seed=0, frames_per_phoneme=4.0, dur_weight_scale=0.25))
    lens = np.array([20, 4, 11])
    inp = wl.synth_inputs(3, 20, seed=2, src_lens=lens)
    host_lens = torch.from_numpy(np.ascontiguousarray(inp[2]))
    m = FastSpeech2Align(wl.preprocess_config("phoneme_level", "frame_level"), dict(cfg, phase1_packing="always")).to("cuda").eval()
    m.load_state_dict(sd)
    with torch.no_grad():
        a = m(dev(inp[0]), dev(inp[1]), host_lens, inp[3])
        assert int(m._lib.ns_last_phase1_rows(m._h)) == 60  # phoneme_level pitch adds its embedding to padded phonemes too: grid
        b = m(dev(inp[0]), dev(inp[1]), dev(inp[2]), inp[3])
    assert all(torch.equal(a[i], b[i]) for i in (0, 1, 2, 3, 4, 5, 9))
    m = FastSpeech2Align(wl.preprocess_config(), dict(cfg, phase1_packing="always")).to("cuda").eval()
    m.load_state_dict(sd)
    uni = wl.synth_inputs(3, 20, seed=2)
    with torch.no_grad():
        m(dev(uni[0]), dev(uni[1]), torch.from_numpy(np.ascontiguousarray(uni[2])), uni[3])
    assert int(m._lib.ns_last_phase1_rows(m._h)) == 60      # uniform lengths: nothing to save
    # the automatic rule: small grids cost steps of 256 workgroups, not rows — the same ragged batch is NOT packed when no step
    # is saved (3 x 20 phonemes), and IS when one is (32 x 128 phonemes at ~0.55 of the rows: 4 steps of the k=9 GEMM -> 3)
    m = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
    m.load_state_dict(sd)
    with torch.no_grad():
        m(dev(inp[0]), dev(inp[1]), host_lens, inp[3])
    assert int(m._lib.ns_last_phase1_rows(m._h)) == 60
    _MODEL.clear()
    cfg, sd = weights_for(dict(config="ljspeech", weight_seed=0, frames_per_phoneme=2.0, dur_weight_scale=0.25))
    m = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
    m.load_state_dict(sd)
    rr = np.random.RandomState(3)
    lens = rr.randint(8, 129, size=32)
    lens[0] = 128
    big = wl.synth_inputs(32, 128, seed=4, src_lens=lens)
    with torch.no_grad():
        o_pk = m(dev(big[0]), dev(big[1]), torch.from_numpy(np.ascontiguousarray(big[2])), big[3])
        rows_auto = int(m._lib.ns_last_phase1_rows(m._h))
        o_gr = m(dev(big[0]), dev(big[1]), dev(big[2]), big[3])
    expect = int(np.minimum(lens + 2, 128).sum())
    steps = lambda r: -(-((r + 31) // 32) * 8 // 256)  # noqa: E731
    assert (rows_auto == expect) == (expect * 10 <= 32 * 128 * 9 and steps(expect) < steps(32 * 128)), (rows_auto, expect)
    assert torch.equal(o_pk[5], o_gr[5]) and torch.equal(o_pk[9], o_gr[9])


def test_empty_utterance_in_a_batch_end_to_end():
    """SURVEY.md §8b "Errors": an utterance with src_len == 0 has every attention key masked — an all -inf softmax row, NaN
    (transformer/Modules.py:19-22) — which FFTBlock's masked_fill then replaces by zeros (transformer/Layers.py:43,46), so it comes out
    with zero durations and zero frames while its neighbours are untouched.  Through the whole forward, against the oracle: same NaN
    pattern (none survives in the returned tensors' valid regions), same integers, same values — on the grid, with phase 1 on packed
    phoneme rows (host src_lens: the empty utterance's window is its two guard rows) and with phase 2 on packed rows."""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import fs2_oracle as orc
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    _MODEL.clear()
    cfg, sd = weights_for(dict(config="tiny", weight_seed=0, frames_per_phoneme=4.0, dur_weight_scale=0.25))
    w = orc.to_torch_weights(sd)
    lens = np.array([20, 0, 7, 13])
    inp = wl.synth_inputs(4, 20, seed=6, src_lens=np.maximum(lens, 1))
    texts = inp[1].copy()
    texts[1, :] = 0   # the empty utterance: all padding tokens
    with torch.no_grad():
        ref = orc.forward(w, cfg, torch.from_numpy(inp[0]), torch.from_numpy(texts), torch.from_numpy(lens), inp[3])
    assert int(ref[9][1]) == 0 and int(ref[9][0]) > 0
    for mode, pack1 in (("dense", "never"), ("packed", "always")):
        m = FastSpeech2Align(wl.preprocess_config(), dict(cfg, padded_rows=mode, phase1_packing=pack1)).to("cuda").eval()
        m.load_state_dict(sd)
        for lens_arg in (dev(lens), torch.from_numpy(lens.copy())):
            with torch.no_grad():
                out = m(dev(inp[0]), dev(texts), lens_arg, inp[3], p_targets=ref[2].cuda(), e_targets=ref[3].cuda())
            torch.cuda.synchronize()
            what = f"{mode}, src_lens on the {'device' if lens_arg.is_cuda else 'host'}"
            assert np.array_equal(out[5].cpu().numpy(), ref[5].numpy()) and np.array_equal(out[9].cpu().numpy(), ref[9].numpy()), what
            assert np.array_equal(out[6].cpu().numpy(), ref[6].numpy()) and np.array_equal(out[7].cpu().numpy(), ref[7].numpy()), what
            for i in (0, 1, 3, 4):
                g, r = out[i].cpu(), ref[i]
                assert torch.equal(torch.isnan(g), torch.isnan(r)), (what, NAMES[i], int(torch.isnan(g).sum()), int(torch.isnan(r).sum()))
                ok = ~torch.isnan(r)
                assert float((g[ok] - r[ok]).abs().max()) < 2e-5, (what, NAMES[i])


@pytest.mark.parametrize("name", ["f64_cfg1", "f64_cfg2", "f64_cfg4", "f64_cfg5"])
def test_accuracy_against_float64(name):
    """Not 'close to the fp32 reference' but 'as close to the TRUTH as the fp32 reference is': the imported reference cast to
    .double() (tests/golden/f64_cfg*.npz, tools/reference_self_deviation.py; bucket decisions pinned to the fp32 reference's own on
    all three sides, so the three evaluations differ in arithmetic only) is the truth, and per quantity the HIP path's p99.9 and max
    distance from it must stay within 1.5x the fp32 reference's own (model/modules.py:80-100,132-135, transformer/SubLayers.py:87-95).
    Round 5's build — one sequential fp32 MFMA sum over the k=9 convolution's 2304 products — measured 2.5-3.75x on pitch / energy
    (profiles/r06_accuracy_vs_f64_sequential.md); the chunked accumulation of csrc/gemm_conv.hip (ACC2) brought every quantity to
    <= 1.1x.  The absolute floor covers quantities where both sides sit at a few ulp (1.5x of 2e-7 is noise, not accuracy)."""
    from tests.util import F64_FIXTURES, accuracy_against_float64

    meta, _ = load_golden(F64_FIXTURES[name])
    cfg, sd, m = gpu_model(meta)
    res = accuracy_against_float64(name, m, sd)
    floor = {"log_d": 4e-7, "pitch_rel": 2e-6, "energy_rel": 2e-6, "mel": 1e-6, "postnet": 1e-6}
    worst = 0.0
    for q, v in res.items():
        for stat in ("p999", "max"):
            h, r = v["hip"][stat], v["ref32"][stat]
            print(f"{name} {q:10s} {stat:5s} |HIP - f64| {h:.3e}   |reference-fp32 - f64| {r:.3e}   ratio {h / max(r, 1e-30):.2f}")
            worst = max(worst, h / max(r, 1e-30))
            assert h <= max(1.5 * r, floor[q]), (name, q, stat, "HIP is further from float64 than 1.5x the fp32 reference", h, r)
    print(f"{name}: worst ratio {worst:.2f}")
    _MODEL.clear()


def test_bits_across_launch_plans():
    """The reference is bit-identical for an utterance whatever its neighbours and whatever T_pad (fixture
    e2e_tiny_neighbours; transformer/Layers.py:43,46 — the blocks are leak-free).  Here a launch picks its tile family from the
    batch's ROW COUNT (csrc/gemm_conv.hip plan_rows: v_mfma 32x32x2 or 16x16x4, which walk k in different orders; a cut plan or one
    launch; packed or dense rows), so the same utterance carries different BITS in batches of different size — within fp32 noise,
    with identical integers.  This test pins that statement: seven utterances (padded in every batch, like F3c's) alone in a batch
    of 8, inside a batch of 9 and inside a batch of 17 (MF 32 <-> 16, one launch <-> cut), with src_lens on the device (dense rows)
    and on the host (packed rows): durations / frame counts identical, bucket decisions pinned, |delta mel| within the bound
    measured on the round-6 build (mel 1.0e-6, PostNet mel 1.1e-6, pitch 6.1e-6 and energy 7.4e-6 relative on the in-range
    frames, log-duration 2.4e-7) x 2 — and replicas INSIDE one batch stay bit-identical
    (test_large_batch_replicas_are_identical).  INTEGRATION.md 'Bits and batch size' states the same for callers."""
    import smart_nar_fast_tts_amd.workload as wl
    from oracle import parity

    meta = dict(config="ljspeech", weight_seed=0, frames_per_phoneme=8.0, dur_weight_scale=0.25)
    cfg, sd, m = gpu_model(meta)
    L = 128
    lens = np.array([128, 96, 128, 77, 120, 128, 101, 64] + [128, 90, 128, 128, 70, 128, 110, 128, 99])
    sp, tx, ln, _ = wl.synth_inputs(17, L, seed=21, src_lens=lens)
    keep = m.packed_rows

    def run(n, host_lens, p_t=None, e_t=None):
        m.packed_rows = host_lens
        lens_t = torch.from_numpy(np.ascontiguousarray(ln[:n])) if host_lens else dev(ln[:n])
        with torch.no_grad():
            return m(dev(sp[:n]), dev(tx[:n]), lens_t, L, p_targets=p_t, e_targets=e_t)

    try:
        base_free = run(8, False)
        T8 = int(base_free[0].shape[1])
        mel_lens = base_free[9].cpu().numpy()
        longest = int(np.argmax(mel_lens))  # un-padded in the batch of 8, padded in the larger ones if a neighbour is longer: left out
        rows = [i for i in range(8) if i != longest]
        base = run(8, False, base_free[2], base_free[3])
        worst = {"mel": 0.0, "postnet": 0.0, "pitch_rel": 0.0, "energy_rel": 0.0, "log_d": 0.0}
        plans = {}
        for n in (8, 9, 17):
            for host in (False, True):
                free = run(n, host)
                T = int(free[0].shape[1])
                assert T >= T8
                # the seven utterances take the batch-of-8 run's bucket decisions; the other rows their own
                p_t, e_t = free[2].clone(), free[3].clone()
                p_t[:8] = 0.0
                e_t[:8] = 0.0
                p_t[:8, :T8] = base_free[2]
                e_t[:8, :T8] = base_free[3]
                if T > T8:  # the batch-of-8's longest gained padding: its own values there (it is not compared)
                    p_t[longest], e_t[longest] = free[2][longest], free[3][longest]
                out = run(n, host, p_t, e_t)
                plans[(n, host)] = (_plan_of(m, n * T, cfg["transformer"]["conv_filter_size"], 256, 9), int(m._lib.ns_last_phase2_rows(m._h)))
                assert torch.equal(out[5][:8], base[5]) and torch.equal(out[9][:8], base[9]), (n, host, "integers differ")
                assert torch.equal(free[5][:8], base[5]), (n, host, "free-running durations differ")
                for i in rows:
                    t = int(mel_lens[i])
                    for key, k in (("mel", 0), ("postnet", 1)):
                        worst[key] = max(worst[key], float((out[k][i, :t] - base[k][i, :t]).abs().max()))
                    # (pitch / energy: on the frames inside the bin range, relative — the quantity a bucket decision hangs on,
                    # oracle/parity.py; near zero the predictor's output is a cancelling sum whose absolute noise is that of its
                    # +-500 summands)
                    for key, k, bins in (("pitch_rel", 2, "variance_adaptor.pitch_bins"), ("energy_rel", 3, "variance_adaptor.energy_bins")):
                        worst[key] = max(worst[key], parity.max_rel_deviation(out[k][i, :t].cpu().numpy(), base[k][i, :t].cpu().numpy(),
                                                                              np.asarray(sd[bins]), np.ones(t, dtype=bool)))
                    s = int(ln[i])
                    worst["log_d"] = max(worst["log_d"], float((out[4][i, :s] - base[4][i, :s]).abs().max()))
        print("launch plans of the decoder k=9 GEMM (rows on the grid, plan) and phase-2 rows:", plans)
        print("the same seven utterances in batches of 8 / 9 / 17, device + dense vs host + packed lengths, worst difference:", worst)
        mfs = {p[0]["mfma_edge"] for p in plans.values() if p[0]}
        assert len({(p[0] or {}).get("main") for p in plans.values()}) > 1, "the three batch sizes were meant to take different plans"
        assert worst["mel"] <= 2.5e-6 and worst["postnet"] <= 2.5e-6 and worst["log_d"] <= 6e-7, worst
        assert worst["pitch_rel"] <= 1.5e-5 and worst["energy_rel"] <= 1.5e-5, worst
        print("MFMA tile edges seen:", mfs)
    finally:
        m.packed_rows = keep


@pytest.mark.parametrize("name", ["pin_cfg1_single", "pin_cfg2_b16", "pin_cfg3_b128_sharded", "pin_cfg4_d512", "pin_cfg5_longform"])
def test_free_running_at_baseline_size(name):
    """north_star's criterion as worded — FREE-RUNNING, nothing pinned — at BASELINE sizes against the reference's own values
    (tests/golden/pin_cfg*.npz): integers identical; every bucket decision that differs from the reference's sits on a bin edge
    (within EDGE_REL, by one bucket); and every utterance in which NO decision differs has its mel and PostNet mel within 1e-3 on
    every stored frame.  An utterance with an edge flip is the case the reference itself produces against itself
    (profiles/r06_reference_self_deviation.md: mkldnn off flips one energy decision of pin_cfg3 and moves 290 frames by up to
    1.07); it is counted and reported, and the bucket-pinned tests above cover it."""
    from oracle import parity

    meta, z = load_golden(name)
    cfg, sd, m = gpu_model(meta)
    out = run_gpu(m, z, meta)
    assert np.array_equal(out[9].cpu().numpy(), z["mel_lens"]) and np.array_equal(out[5].cpu().numpy(), z["d_rounded"])
    valid = ~z["mel_masks"]
    pb, eb = np.asarray(sd["variance_adaptor.pitch_bins"]), np.asarray(sd["variance_adaptor.energy_bins"])
    hit = np.zeros(valid.shape[0], dtype=bool)
    n_flip = 0
    for i, key, bins, edge in ((2, "p_predictions", pb, "p_edge_rel"), (3, "e_predictions", eb, "e_edge_rel")):
        got, ref = parity.bucketize(out[i].cpu().numpy(), bins), parity.bucketize(z[key], bins)
        flip = (got != ref) & valid
        if key[0] == "p":
            p_hit = flip.any(axis=1)
        hit |= flip.any(axis=1)
        n_flip += int(flip.sum())
        if key[0] == "p":  # (an energy decision downstream of a flipped pitch decision is a consequence, not a flip of its own)
            assert np.all(z[edge][flip] < EDGE_REL) and np.all(np.abs(got - ref)[flip] <= 1), (name, key, "flip away from an edge")
        else:
            own = flip & ~p_hit[:, None]
            assert np.all(z[edge][own] < EDGE_REL) and np.all(np.abs(got - ref)[own] <= 1), (name, key, "flip away from an edge")
    stride = int(meta["frame_stride"])
    clean = ~hit
    vs = valid[:, ::stride] & clean[:, None]
    worst = 0.0
    for i, key in ((0, "output_sub"), (1, "postnet_output_sub")):
        err = np.abs(out[i].cpu().numpy()[:, ::stride].astype(np.float64) - z[key])[vs]
        worst = max(worst, float(err.max()) if err.size else 0.0)
    print(f"{name}: free-running, {int(clean.sum())} of {len(clean)} utterances without a differing decision: worst mel error {worst:.2e}; "
          f"{n_flip} decisions differ (all on an edge) in {int(hit.sum())} utterance(s)")
    assert worst < MEL_TOL, (name, worst)
    assert clean.sum() >= len(clean) - 2, (name, "more than two utterances hit a bin edge", int(hit.sum()))
    _MODEL.clear()
