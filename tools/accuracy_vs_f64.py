#!/usr/bin/env python3
"""How far the HIP path is from a float64 evaluation of the same forward, beside the fp32 reference's own distance — GPU box.

    gpurun -- 'python tools/accuracy_vs_f64.py r06 [--matmul bf16x3] [--ops]'  ->  gpurun_out/r06_accuracy_vs_f64[_bf16x3].{md,json}

(1) End to end, per quantity, on the float64 fixtures tests/golden/f64_cfg{1,2,4,5}.npz (the imported reference cast to
    .double(), written in the build container by tools/reference_self_deviation.py): max / p99.9 / median of |HIP - f64| and of
    |reference-fp32 - f64|, bucket decisions pinned to the fp32 reference's on all three sides.  This is the table
    tests/test_gpu_parity.py::test_accuracy_against_float64 asserts on.
(2) ``--ops``: operator by operator on identical inputs (the oracle evaluated in float64 and in fp32 on this box's host, the HIP
    operator on the device): which operator, if any, is further from the truth than torch's CPU kernels.

TEST INFRASTRUCTURE: reads tests/golden and imports oracle/ as the checker; nothing here is on the product path.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import F64_FIXTURES, accuracy_against_float64, load_golden, weights_for  # noqa: E402

import smart_nar_fast_tts_amd.workload as wl  # noqa: E402
from smart_nar_fast_tts_amd.model import FastSpeech2Align  # noqa: E402


def st(d):
    d = np.abs(np.asarray(d, dtype=np.float64)).reshape(-1)
    return {"max": float(d.max()), "p999": float(np.quantile(d, 0.999)), "rms": float(np.sqrt((d ** 2).mean()))}


def per_op(model, cfg, sd, B=4, S=640, seed=0):
    """Each operator on the same random inputs: HIP vs oracle-fp32 vs oracle-f64 (errors absolute; activations are O(1))."""
    from oracle import fs2_oracle as orc
    from smart_nar_fast_tts_amd import ops

    w32 = orc.to_torch_weights(sd)
    w64 = {k: (v.double() if v.is_floating_point() else v) for k, v in w32.items()}
    rs = np.random.RandomState(seed)
    d, nh = cfg["transformer"]["decoder_hidden"], cfg["transformer"]["decoder_head"]
    x = torch.from_numpy(rs.standard_normal((B, S, d)).astype(np.float32))
    lens = torch.tensor([S] + [int(v) for v in rs.randint(S // 2, S, size=B - 1)])
    mask = orc.get_mask_from_lengths(lens, S)
    xm = x.masked_fill(mask.unsqueeze(-1), 0.0)
    mel = torch.from_numpy(rs.standard_normal((B, S, 80)).astype(np.float32))
    xd, ld = x.cuda(), lens.cuda()
    p = "mel_decoder.layer_stack.0"
    rows = {}

    def rec(name, hip, f32, f64, sel=None):
        h, a, t = hip.cpu().numpy().astype(np.float64), f32.numpy().astype(np.float64), f64.numpy()
        if sel is not None:
            h, a, t = h[sel], a[sel], t[sel]
        rows[name] = {"hip": st(h - t), "torch_cpu_fp32": st(a - t), "scale_rms": float(np.sqrt((t ** 2).mean()))}

    with torch.no_grad():
        k1 = w32[p + ".pos_ffn.w_1.weight"].shape[2]
        conv = lambda w, t: torch.relu(torch.nn.functional.conv1d(t.transpose(1, 2), w[p + ".pos_ffn.w_1.weight"], w[p + ".pos_ffn.w_1.bias"], padding=(k1 - 1) // 2)).transpose(1, 2)  # noqa: E731
        rec("ffn w_1 (k=9 conv, K=%d) + ReLU" % (d * k1), ops.ffn_conv1(model, p + ".pos_ffn", xd), conv(w32, x), conv(w64, x.double()))
        rec("positionwise_ffn (conv9 -> conv1 -> +x -> LN)", ops.positionwise_ffn(model, p + ".pos_ffn", xd), orc.positionwise_ffn(w32, p + ".pos_ffn", x),
            orc.positionwise_ffn(w64, p + ".pos_ffn", x.double()))
        valid = ~mask.numpy()
        rec("multi_head_attention (QKV -> softmax -> fc -> +x -> LN)", ops.multi_head_attention(model, p + ".slf_attn", xd, ld),
            orc.multi_head_attention(w32, p + ".slf_attn", x, mask, nh), orc.multi_head_attention(w64, p + ".slf_attn", x.double(), mask, nh))
        rec("fft_block", ops.fft_block(model, p, xd, ld), orc.fft_block(w32, p, x, mask, nh), orc.fft_block(w64, p, x.double(), mask, nh))
        rec("mel_decoder (pos + %d blocks)" % cfg["transformer"]["decoder_layer"], ops.mel_decoder(model, xm.cuda(), ld),
            orc.mel_decoder(w32, xm, mask, nh, cfg["max_seq_len"]), orc.mel_decoder(w64, xm.double(), mask, nh, cfg["max_seq_len"]), valid)
        vp = "variance_adaptor.pitch_predictor"
        rec("variance_predictor (pitch)", ops.variance_predictor(model, vp, xd, ld), orc.variance_predictor(w32, vp, x, mask),
            orc.variance_predictor(w64, vp, x.double(), mask), valid)
        rec("mel_linear", ops.mel_linear(model, xd), torch.nn.functional.linear(x, w32["mel_linear.weight"], w32["mel_linear.bias"]),
            torch.nn.functional.linear(x.double(), w64["mel_linear.weight"], w64["mel_linear.bias"]))
        rec("postnet (5 x conv5, BN folded, tanh)", ops.postnet(model, mel.cuda()), orc.postnet(w32, mel), orc.postnet(w64, mel.double()))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("round", nargs="?", default="r06")
    ap.add_argument("--matmul", default="fp32", choices=["fp32", "bf16x3"])
    ap.add_argument("--ops", action="store_true")
    ap.add_argument("--fixtures", nargs="+", default=list(F64_FIXTURES))
    args = ap.parse_args()
    rep = {"round": args.round, "matmul": args.matmul, "device": torch.cuda.get_device_name(0), "fixtures": {}, "ops": {}}
    for name in args.fixtures:
        meta, _ = load_golden(F64_FIXTURES[name])
        cfg, sd = weights_for(meta)
        cfg = dict(cfg)
        if args.matmul == "bf16x3":
            cfg["matmul"] = "bf16x3"
        m = FastSpeech2Align(wl.preprocess_config(), cfg).to("cuda").eval()
        m.load_state_dict(sd)
        rep["fixtures"][name] = accuracy_against_float64(name, m, sd)
        print(name, json.dumps(rep["fixtures"][name]), flush=True)
        if args.ops and name in ("f64_cfg2", "f64_cfg4"):
            rep["ops"][name] = per_op(m, cfg, sd)
            print(name, "ops", json.dumps(rep["ops"][name]), flush=True)
        del m
        torch.cuda.empty_cache()
    L = [f"# Distance from a float64 evaluation of the forward — {args.round}, matmul = {args.matmul}", "",
         "`tools/accuracy_vs_f64.py` on the MI355X box.  Truth: the imported reference cast to `.double()` (tests/golden/f64_cfg*.npz, "
         "`tools/reference_self_deviation.py`).  All three evaluations take the fp32 reference's bucket decisions, so they differ in arithmetic only.",
         "pitch / energy: relative to max(|truth|, 1), frames inside the bin range; log_d and mels absolute (mels: every 16th frame).", "",
         "| fixture | quantity | HIP max | reference-fp32 max | ratio | HIP p99.9 | reference-fp32 p99.9 | ratio | HIP median | reference-fp32 median |",
         "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for name, r in rep["fixtures"].items():
        for q, v in r.items():
            h, a = v["hip"], v["ref32"]
            L.append(f"| {name} | {q} | {h['max']:.2e} | {a['max']:.2e} | {h['max'] / max(a['max'], 1e-30):.2f} | {h['p999']:.2e} | {a['p999']:.2e} | "
                     f"{h['p999'] / max(a['p999'], 1e-30):.2f} | {h['median']:.2e} | {a['median']:.2e} |")
    if rep["ops"]:
        L += ["", "## Operator by operator, identical inputs (B = 4, S = 640, N(0,1) activations; absolute errors against the float64 oracle)", "",
              "| weights | operator | HIP rms | torch-CPU-fp32 rms | ratio | HIP max | torch-CPU-fp32 max | output rms |", "|---|---|---:|---:|---:|---:|---:|---:|"]
        for name, rows in rep["ops"].items():
            for op, v in rows.items():
                h, a = v["hip"], v["torch_cpu_fp32"]
                L.append(f"| {name} | {op} | {h['rms']:.2e} | {a['rms']:.2e} | {h['rms'] / max(a['rms'], 1e-30):.2f} | {h['max']:.2e} | {a['max']:.2e} | {v['scale_rms']:.2e} |")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    base = os.path.join(ROOT, "gpurun_out", f"{args.round}_accuracy_vs_f64" + ("" if args.matmul == "fp32" else "_" + args.matmul))
    json.dump(rep, open(base + ".json", "w"), indent=1)
    open(base + ".md", "w").write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main()
