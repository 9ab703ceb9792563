#!/usr/bin/env python3
"""bench.py — whole-job valid mel-frames/sec of the FastSpeech2 inference forward on N MI355X.

    python bench.py --gpus N --steps 20 --warmup 5          (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one forward() of the hot path over one synthetic utterance batch already resident in HBM
(BASELINE.json config 2: batch 16, phoneme_len 128, mel_len ~1024, d_model 256, 4+4 FFT layers).  With N > 1
every rank runs its own 16-utterance shard (config 3 = 128 utterances over 8 GPUs; weak scaling, no data-path
collective; weights replicated by ONE RCCL broadcast before the timed region).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def rank_env(env) -> None:
    """Environment every rank of a multi-process run needs, set BEFORE torch (and with it HIP / RCCL) is imported — in the
    ranks themselves, so that it holds for BOTH launch forms (`python bench.py --gpus N` re-executing itself, and the
    driver's `python -m torch.distributed.run ... bench.py`, which never passes through self_launch):
    * HSA_ENABLE_IPC_MODE_LEGACY=0: this host driver only supports dmabuf IPC; without it RCCL init / cross-process device
      memory fails with `hipIpcGetMemHandle: invalid argument`;
    * OMP_NUM_THREADS: N ranks x (all host cores) torch-CPU threads would oversubscribe the host (rank 0 also runs the
      CPU-oracle leg at N = 1, where the default stays untouched)."""
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if int(env.get("WORLD_SIZE", "1")) > 1:
        env.setdefault("OMP_NUM_THREADS", "8")


rank_env(os.environ)

import numpy as np  # noqa: E402
import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix), v_mfma_f32_32x32x2_f32
HBM_PEAK_GBS = 8000.0
# the same two ceilings measured on an MI355X box with microbenchmarks (tools/lab/mfma_peak.hip, tools/lab/dma_fill.hip)
F32_MFMA_MEASURED_TFLOPS = 155.0
HBM_MEASURED_GBS = 6400.0
# SURVEY.md §8(d) algorithmic bytes per valid frame (weights once per forward + every sub-layer boundary tensor once each way)
ALGORITHMIC_KB_PER_FRAME = {"cfg1_single": 190.0, "cfg2_b16": 77.0, "cfg3_b128_sharded": 77.0, "cfg4_d512": 174.0,
                            "cfg5_longform": 70.0, "cfg5_longform_gaussian": 70.0}


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: run the same command line as N ranks under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1) and hand its output and exit code through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (the ranks set it themselves too: rank_env)
    env.setdefault("OMP_NUM_THREADS", "8")
    env["NS_BENCH_LAUNCHER"] = "self"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2_b16", help="cfg2_b16 | cfg4_d512 | cfg5_longform | cfg5_longform_gaussian | cfg1_single")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="profiling runs: only warm-up + timed steps (no latency leg, no pipelined leg, no CPU baseline), so that a "
                         "kernel trace holds exactly (warmup + steps) forwards of the workload")
    ap.add_argument("--ragged", action="store_true", help="ragged utterance lengths instead of the uniform BASELINE batch (phase 2 then runs on packed rows; NS_PACKED=0 keeps the padded grid)")
    ap.add_argument("--batch", type=int, default=0, help="override the workload's per-GPU batch size (sweeps; not a BASELINE config)")
    ap.add_argument("--streams", type=int, default=1, help="issue consecutive steps round-robin on this many HIP streams")
    ap.add_argument("--global-pad", action="store_true", help="pad every shard to the global max mel length (all-reduce MAX)")
    ap.add_argument("--matmul", choices=["fp32", "bf16x3"], default="fp32",
                    help="EXPERIMENT, never the headline: bf16x3 runs the large decoder-FFN / PostNet contractions from an exact "
                         "3-way bf16 split on the bf16 matrix cores (fp32-sized error, different bits); the line is labelled")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd import sharding
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # NS_BENCH_ONE_GPU=1 (test rigs only): every rank on cuda:0 with gloo, to exercise the N > 1 flow on a one-GPU box.
    # The JSON says so (one_gpu_rig / devices / backend): such a line is a plumbing check, never a scaling number.
    one_gpu = os.environ.get("NS_BENCH_ONE_GPU") == "1"
    n_dev = torch.cuda.device_count()
    if not one_gpu and local_rank >= n_dev:
        raise SystemExit(f"--gpus {args.gpus} but this box shows {n_dev} GPU(s): rank {rank} has no device "
                         "(NS_BENCH_ONE_GPU=1 runs every rank on cuda:0 as a plumbing check)")
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist, backend = None, None
    # NS_BENCH_FORCE_DIST=1 (test rigs): build the process group even for one rank, so that the RCCL code path (init,
    # weight broadcast, all-reduce, all-gather) is exercised on a one-GPU box
    if world > 1 or os.environ.get("NS_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if one_gpu else "nccl"  # "nccl" is RCCL on ROCm
        if one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    cfg_name, B_shard, L, fpp = wl.WORKLOADS[args.workload]
    if args.batch > 0:
        B_shard = args.batch
    cfg = wl.model_config(cfg_name)
    b3 = args.matmul == "bf16x3"
    if b3:
        cfg["matmul"] = "bf16x3"
    model = FastSpeech2Align(wl.preprocess_config(), cfg).to(dev).eval()
    sd = wl.synth_state_dict(cfg, seed=0, frames_per_phoneme=fpp) if rank == 0 else None
    sharding.broadcast_weights(model, sd, src=0)  # N == 1: plain load_state_dict

    # each rank's shard of the global batch (B_shard utterances per GPU): rows [rank*B, (rank+1)*B) of one seeded batch
    ragged = None
    if args.ragged:  # phoneme counts uniform in [L/8, L], one utterance per shard at L: what unbucketed serving batches look like
        rr = np.random.RandomState(7)
        ragged = rr.randint(max(1, L // 8), L + 1, size=B_shard * world)
        ragged[::B_shard] = L
    sp, tx, ln, _ = wl.synth_inputs(B_shard * world, L, seed=0, src_lens=ragged)
    sp, tx, ln, Lmax = sharding.shard_batch(sp, tx, ln, world, rank)
    speakers, texts, src_lens = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (sp, tx, ln))
    pad_fn = sharding.global_max if (args.global_pad and world > 1) else None

    # --streams S > 1: consecutive steps go round-robin onto S HIP streams, so the small-grid phase 1 of step i+1 (and
    # the host read of mel_lens between the phases) overlaps the chip-filling phase 2 of step i.  Same K steps, same work.
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None
    counter = [0]

    def step():
        if streams is None:
            return model(speakers, texts, src_lens, Lmax, max_mel_len=pad_fn)
        st = streams[counter[0] % len(streams)]
        counter[0] += 1
        with torch.cuda.stream(st):
            return model(speakers, texts, src_lens, Lmax, max_mel_len=pad_fn)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = step()
        fence()
        model.profile_dominant_kernel(True)
        # per-step spread without extra syncs: one event per step on the launch stream, read after the closing fence
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if streams is None else None
        t0 = time.perf_counter()
        if marks:
            marks[0].record()
        for i in range(args.steps):
            out = step()
            if marks:
                marks[i + 1].record()
        fence()
        elapsed = time.perf_counter() - t0
        step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)) if marks else None
        k_ms, k_flops, k_launches = model.read_profile(0)
        model.profile_dominant_kernel(False)
        # the next two heaviest kernels (attention, PostNet 512->512) in a pass of their own, outside the timed region: a
        # timed launch costs its stream ~5 us (include/nar_fs2.h), and seven more of them per forward were 1 % of the step
        by_kernel = {}
        if not args.no_extras:
            model.profile_slots((1, 2))
            for _ in range(min(args.steps, 5)):
                step()
            fence()
            by_kernel = {name: model.read_profile(i) for i, name in enumerate(model.PROFILE_SLOTS) if i > 0}
            model.profile_slots(())

        # Secondary, outside the timed region above: the same K steps issued round-robin on two HIP streams (what
        # batching.synthesize(streams=2) does for consecutive batches), so the small-grid phase 1 and the host read of
        # step i+1 overlap the chip-filling phase 2 of step i.  Reported beside the headline value, never as it.
        pipelined = None
        if streams is None and args.gpus == 1 and not args.no_extras:
            ps = [torch.cuda.Stream(device=dev) for _ in range(2)]
            for i in range(2):
                with torch.cuda.stream(ps[i]):
                    model(speakers, texts, src_lens, Lmax, max_mel_len=pad_fn)
            torch.cuda.synchronize()
            t0p = time.perf_counter()
            for i in range(args.steps):
                with torch.cuda.stream(ps[i % 2]):
                    model(speakers, texts, src_lens, Lmax, max_mel_len=pad_fn)
            torch.cuda.synchronize()
            pipelined = time.perf_counter() - t0p

    frames = int(out[9].sum().item())  # valid frames of this rank's shard (sum of mel_lens, never B*T_pad)
    T_pad = int(out[0].shape[1])
    stats = torch.tensor([elapsed, float(frames), float(T_pad)], dtype=torch.float64, device=dev)
    devices = [{"rank": rank, "device": f"cuda:{dev_index}", "name": torch.cuda.get_device_name(dev)}]
    # per-rank view (SURVEY.md §8e "scaling risks": per-shard T_pad differs, so load imbalance must be visible in a SCALE line)
    per_rank = [{"rank": rank, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "T_pad": T_pad, "valid_frames": frames,
                 "rows_phase2": int(model._lib.ns_last_phase2_rows(model._h)),  # B*T_pad on the grid, fewer on packed rows (ragged batches)
                 "hsa_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "omp_num_threads": os.environ.get("OMP_NUM_THREADS"),
                 "launcher": os.environ.get("NS_BENCH_LAUNCHER", "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else "none")}]
    world_seen = 1
    if dist is not None:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stats.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max, frames_total, T_pad_max = float(tmax[0]), float(tsum[1]), int(tmax[2])
        world_seen = dist.get_world_size()  # what the process group (RCCL on the GPU box) itself reports
        gathered = [None] * world
        dist.all_gather_object(gathered, (devices[0], per_rank[0]))
        devices = [g[0] for g in gathered]
        per_rank = [g[1] for g in gathered]
    else:
        elapsed_max, frames_total, T_pad_max = elapsed, float(frames), T_pad

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = frames_total * args.steps / elapsed_max
    flops_frame = wl.algorithmic_flops_per_frame(cfg, T_pad, L, fpp)
    achieved_tflops = (k_flops / (k_ms * 1e-3)) / 1e12 if k_ms > 0 else 0.0
    # roofline.traffic: HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/collect_profiles.sh)
    # — only when that record was measured on THIS launch geometry (rows = B*T_pad, widths, kernel size); otherwise null
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
    t = cfg["transformer"]
    geom = {"rows": int(out[0].shape[0]) * T_pad, "d_model": t["decoder_hidden"], "d_inner": t["conv_filter_size"],
            "k": t["conv_kernel_size"][0]}
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath))
            if all(rec.get(k) == v for k, v in geom.items()):
                traffic = rec.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    res = {
        "metric": "mel_frames_per_sec", "value": value, "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16x3" if b3 else "f32", "data": "synthetic",
        "devices": devices, "backend": backend, "world_size_seen_by_rccl": world_seen, "one_gpu_rig": one_gpu, "per_rank": per_rank,
        "config": {"workload": f"{args.workload}{'_bf16x3' if b3 else ''}{' (ragged lengths)' if args.ragged else ''}{f' (batch overridden: {args.batch})' if args.batch > 0 else ''}: LJSpeech config, batch {B_shard}/GPU x {args.gpus} GPU, phoneme_len {L}, "
                               f"T_pad {T_pad_max}, d_model {cfg['transformer']['decoder_hidden']}, "
                               f"{cfg['transformer']['encoder_layer']}+{cfg['transformer']['decoder_layer']} FFT layers, "
                               f"random-init weights (seed 0, duration bias log({fpp + 1:g}))",
                   "global_batch": B_shard * args.gpus, "valid_frames_per_step": int(frames_total),
                   "padding": "global-pad" if pad_fn else "per-shard",
                   "algorithmic_mflop_per_frame": round(flops_frame / 1e6, 2),
                   "end_to_end_tflops": round(flops_frame * value / 1e12, 2)},
        "roofline": {"bound": "mfma", "kernel": "k_conv_gemm (FFN w_1: Conv1d k=9, d->d_inner, bias+ReLU)",
                     "achieved": round(achieved_tflops, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved_tflops / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                     # algorithmic bytes of one launch: activation rows read once, the weight matrix once, the output written once
                     "algorithmic_bytes": (geom["rows"] * geom["d_model"] + geom["d_inner"] * geom["k"] * geom["d_model"] + geom["rows"] * geom["d_inner"]) * 4,
                     "traffic_over_algorithmic": None if traffic is None else round(
                         traffic / ((geom["rows"] * geom["d_model"] + geom["d_inner"] * geom["k"] * geom["d_model"] + geom["rows"] * geom["d_inner"]) * 4), 3),
                     "frac_of_measured_peak": round(achieved_tflops / F32_MFMA_MEASURED_TFLOPS, 4),
                     "launches": int(k_launches), "avg_launch_ms": round(k_ms / max(k_launches, 1), 4),
                     "share_of_step_time": round((k_ms * 1e-3) / elapsed if elapsed > 0 else 0.0, 3),
                     "geometry": geom},
    }
    if b3:
        # EXPERIMENT line: the timed kernels ran on the bf16 matrix cores, 6 bf16 MFMA products per fp32 product; price them
        # against the dense bf16 peak (2.5 PFLOP/s, MI355X_MICROARCH.md) by the bf16 flops they EXECUTE
        ro = res["roofline"]
        ro.update({"kernel": "k_conv_gemm_b3 (FFN w_1 from an exact 3-way bf16 split, 6 products, fp32 accumulate)",
                   "fp32_equivalent_tflops": ro["achieved"], "achieved": round(6 * achieved_tflops, 1), "peak": 2500.0,
                   "frac": round(6 * achieved_tflops / 2500.0, 4), "traffic": None,
                   "note": "achieved = 6 x algorithmic fp32 flops / time = bf16 MFMA flops executed; peak = dense bf16 MFMA"})
        ro.pop("frac_of_measured_peak", None)
        res["experiment"] = ("opt-in precision mode, NOT the reference's arithmetic: operands are split exactly into three bf16 pieces; "
                             "results differ from the fp32 path in the last bits (same error size vs fp64). The headline line is "
                             "`python bench.py` without --matmul.")
    # the next two heaviest kernels, timed by the same in-forward HIP events (rank 0's shard): fused attention of the
    # decoder stack (4*rows*T_pad*d flop per launch) and the PostNet's 512->512 k=5 convolutions
    kdesc = {"attention": "k_attention (decoder self-attention: QK^T, key-mask, online softmax, PV)",
             "postnet_mid": "k_conv_gemm (PostNet Conv1d k=5 512->512 + folded BatchNorm + tanh)"}
    res["roofline_by_kernel"] = {"ffn_w1": {k: res["roofline"][k] for k in ("achieved", "frac", "launches", "avg_launch_ms",
                                                                           "share_of_step_time")}}
    for name, (ms, fl, n) in by_kernel.items():
        tf = (fl / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0
        res["roofline_by_kernel"][name] = {"kernel": kdesc.get(name, name), "bound": "mfma", "achieved": round(tf, 2),
                                           "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4),
                                           "launches": int(n), "avg_launch_ms": round(ms / max(n, 1), 4),
                                           "share_of_step_time": round((ms / max(n, 1)) * (n / min(args.steps, 5)) / (elapsed / args.steps * 1e3), 3) if elapsed > 0 and n else 0.0,
                                           "timed_in": "separate pass after the timed region"}

    # whole forward against both ceilings (MFMA primary, HBM secondary; vendor and measured peaks), per GPU
    kb = ALGORITHMIC_KB_PER_FRAME.get(args.workload)
    e2e_tf = flops_frame * value / 1e12 / args.gpus
    res["end_to_end"] = {"tflops_per_gpu": round(e2e_tf, 2), "frac_mfma_peak": round(e2e_tf / F32_MFMA_PEAK_TFLOPS, 4),
                         "frac_mfma_measured": round(e2e_tf / F32_MFMA_MEASURED_TFLOPS, 4),
                         "algorithmic_kb_per_frame": kb,
                         "hbm_gbs_per_gpu": None if kb is None else round(kb * 1e3 * value / args.gpus / 1e9, 1),
                         "frac_hbm_peak": None if kb is None else round(kb * 1e3 * value / args.gpus / 1e9 / HBM_PEAK_GBS, 4),
                         "frac_hbm_measured": None if kb is None else round(kb * 1e3 * value / args.gpus / 1e9 / HBM_MEASURED_GBS, 4)}

    if pipelined:
        res["pipelined"] = {"streams": 2, "steps": args.steps, "ms_per_step": round(pipelined / args.steps * 1e3, 4),
                            "value": round(frames_total * args.steps / pipelined, 1), "unit": "frames/s",
                            "note": "consecutive steps round-robin on 2 HIP streams, measured after the timed region; not the headline value"}
    if step_ms:
        res["step_ms_spread"] = {"p50": round(step_ms[len(step_ms) // 2], 3), "min": round(step_ms[0], 3),
                                 "max": round(step_ms[-1], 3), "n": len(step_ms)}
    if args.gpus == 1 and not args.no_extras:
        # p50 per-utterance latency (the second half of BASELINE.json's metric), config 1: B=1, L=100
        c1, B1, L1, f1 = wl.WORKLOADS["cfg1_single"]
        if c1 == cfg_name and f1 == fpp:  # same architecture AND same synthetic duration bias (mel_len 788)
            s1, t1, l1, _ = wl.synth_inputs(B1, L1, seed=0)
            a1 = [torch.from_numpy(a).to(dev) for a in (s1, t1, l1)]
            lat = []
            with torch.no_grad():
                for i in range(25):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    o1 = model(a1[0], a1[1], a1[2], L1)
                    torch.cuda.synchronize()
                    if i >= 5:
                        lat.append((time.perf_counter() - t0) * 1e3)
            res["latency"] = {"workload": "cfg1_single: B=1, phoneme_len 100", "p50_ms": round(float(np.median(lat)), 3),
                              "min_ms": round(min(lat), 3), "max_ms": round(max(lat), 3), "mel_len": int(o1[9][0]), "n": len(lat)}
            # the same utterance in CAPACITY MODE (forward(max_mel_len=<int>, async_status=True), model/modules.py:128-131 `max_len` semantics): the
            # caller fixes the mel axis, so nothing on the host waits for mel_lens between the two phases.  Capacity = the
            # utterance's own length here, so the device work is identical to the run above (and so are the outputs, bit for bit).
            cap = int(o1[9][0])
            latc = []
            with torch.no_grad():
                for i in range(25):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    oc = model(a1[0], a1[1], a1[2], L1, max_mel_len=cap, async_status=True)
                    torch.cuda.synchronize()
                    if i >= 5:
                        latc.append((time.perf_counter() - t0) * 1e3)
            res["latency_capacity_mode"] = {"workload": f"cfg1_single with max_mel_len={cap}, async_status=True (no host read between the phases)",
                                            "p50_ms": round(float(np.median(latc)), 3), "min_ms": round(min(latc), 3),
                                            "max_ms": round(max(latc), 3), "n": len(latc), "status": oc.check(),
                                            "bit_identical_to_sync_path": bool(torch.equal(oc[1], o1[1]) and torch.equal(oc[9], o1[9]))}

    if args.gpus == 1 and not args.no_extras and not b3:
        # Variable-length batches (BASELINE.json config 5: "variable-length masking stress"): the same model on a RAGGED batch of
        # this workload's shape — phoneme counts uniform in [L/8, L], one utterance at L — with phase 2 on packed rows
        # (include/nar_fs2.h ns_forward_mel_packed, the synchronous path's default) and on the reference's padded [B, T] grid.
        # A secondary figure: the headline value above is BASELINE.json's uniform batch.
        rr = np.random.RandomState(7)
        rl = rr.randint(max(1, L // 8), L + 1, size=B_shard)
        rl[0] = L
        rs_, rt_, rln_, _ = wl.synth_inputs(B_shard, L, seed=0, src_lens=rl)
        ra = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (rs_, rt_, rln_)]
        vl = {}
        keep = model.packed_rows
        try:
            for mode in ("packed", "grid"):
                model.packed_rows = mode == "packed"
                with torch.no_grad():
                    for _ in range(3):
                        ro = model(ra[0], ra[1], ra[2], L)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        ro = model(ra[0], ra[1], ra[2], L)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / 10
                vf = int(ro[9].sum())
                vl[mode] = {"ms_per_step": round(dt * 1e3, 3), "value": round(vf / dt, 1), "unit": "frames/s",
                            "rows_phase2": int(model._lib.ns_last_phase2_rows(model._h))}
                vl["T_pad"], vl["valid_frames"] = int(ro[0].shape[1]), vf
        finally:
            model.packed_rows = keep
        vl["speedup"] = round(vl["grid"]["ms_per_step"] / vl["packed"]["ms_per_step"], 3)
        vl["workload"] = f"{args.workload} with ragged lengths (phoneme counts uniform in [L/8, L], B={B_shard})"
        res["variable_length"] = vl

    if args.gpus == 1 and not args.no_cpu_baseline and not args.no_extras:
        # CPU baseline beside it: the oracle (a torch-CPU restatement of the reference forward, "port") on this box's
        # host cores, same workload, bounded to ~10-30 s.
        from oracle import fs2_oracle as orc

        cores = min(os.cpu_count() or 1, 64)
        torch.set_num_threads(cores)
        w = orc.to_torch_weights(wl.synth_state_dict(cfg, seed=0, frames_per_phoneme=fpp))
        ci = [torch.from_numpy(np.ascontiguousarray(a)) for a in (sp, tx, ln)]
        lr = cfg.get("length_regulator", "hard")
        with torch.no_grad():
            ref = orc.forward(w, cfg, ci[0], ci[1], ci[2], Lmax, length_regulator=lr)  # warm-up
            n, t0 = 0, time.perf_counter()
            while n < 5 and (time.perf_counter() - t0) < 20.0:
                ref = orc.forward(w, cfg, ci[0], ci[1], ci[2], Lmax, length_regulator=lr)
                n += 1
            cpu_t = (time.perf_counter() - t0) / n
        cpu_frames = orc.valid_frames(ref)
        res["cpu_baseline"] = {"value": cpu_frames / cpu_t, "unit": "frames/s", "cores": cores, "kind": "port",
                               "sample": f"{n} forward(s) of the same workload ({args.workload}, {cpu_frames} valid frames each), "
                                         f"{cpu_t:.2f} s per forward, torch {torch.__version__} CPU kernels, {cores} threads"}
        # The checker beside the measurement: same inputs through the HIP path and the oracle.  Free-running, the two may
        # pick different pitch/energy buckets for values that sit on a bucket edge (fp32 summation order);
        # with the oracle's pitch/energy handed to the HIP path as p_targets/e_targets the discrete choices are pinned.
        # What the figures mean: durations / frame counts must be identical.  Free-running, any fp32 evaluation other than
        # torch-CPU's may legally pick the neighbouring embedding row for a value within EDGE_REL (2e-5 relative) of a bin
        # edge (model/modules.py:86-88,97-99); one such flip changes every frame of its utterance through global attention,
        # which is what postnet_max_abs_free_running shows.  bucket_flips counts them, bucket_flips_off_edge counts flips
        # that are NOT at an edge (must be 0: that would be a real error), and the pinned run is the parity number.
        from oracle import parity

        chk = {"postnet_max_abs_free_running": None, "postnet_max_abs_buckets_pinned": None, "durations_equal": None}
        chk["durations_equal"] = bool(torch.equal(out[5].cpu(), ref[5]))
        chk["duration_flips"] = int((out[5].cpu() != ref[5]).sum())
        chk["frame_counts_equal"] = bool(torch.equal(out[9].cpu(), ref[9]))
        if out[1].shape == ref[1].shape:
            valid = ~ref[7].numpy()
            pbins, ebins = w["variance_adaptor.pitch_bins"].numpy(), w["variance_adaptor.energy_bins"].numpy()
            diff = (out[1].cpu() - ref[1]).abs()
            chk["postnet_max_abs_free_running"] = float(diff.max())
            chk["frames_over_1e-3_free_running"] = int(((diff.amax(dim=2) > 1e-3).numpy() & valid).sum())
            chk["valid_frames"] = int(valid.sum())
            with torch.no_grad():
                # energy is predicted on x + pitch_embedding: classify its buckets with the pitch decisions pinned
                pin_p = model(speakers, texts, src_lens, Lmax, p_targets=ref[2].to(dev))
                pin = model(speakers, texts, src_lens, Lmax, p_targets=ref[2].to(dev), e_targets=ref[3].to(dev))
            fp = parity.classify_bucket_flips(out[2].cpu().numpy(), ref[2].numpy(), pbins, valid)
            fe = parity.classify_bucket_flips(pin_p[3].cpu().numpy(), ref[3].numpy(), ebins, valid)
            chk["bucket_decisions"] = 2 * int(valid.sum())
            chk["bucket_flips"] = fp[0] + fe[0]
            chk["bucket_flips_off_edge"] = fp[1] + fe[1]
            chk["bucket_flips_by_more_than_one"] = fp[2] + fe[2]
            chk["edge_rel_bound"] = parity.EDGE_REL
            # the implementation's own distance to the reference on every frame where a decision can change (the bound above is
            # 2x the worst value of this quantity over the five pins, profiles/r03_bucket_edge_deviation.md)
            chk["pitch_max_rel_dev_in_range"] = parity.max_rel_deviation(out[2].cpu().numpy(), ref[2].numpy(), pbins, valid)
            chk["energy_max_rel_dev_in_range"] = parity.max_rel_deviation(pin_p[3].cpu().numpy(), ref[3].numpy(), ebins, valid)
            chk["pitch_max_abs"] = float((out[2].cpu() - ref[2]).abs().max())
            chk["postnet_max_abs_buckets_pinned"] = float((pin[1].cpu() - ref[1]).abs().max())
            chk["frames_over_1e-3_buckets_pinned"] = int((((pin[1].cpu() - ref[1]).abs().amax(dim=2) > 1e-3).numpy() & valid).sum())
        res["check_vs_oracle"] = chk
    print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
