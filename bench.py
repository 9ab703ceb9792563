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


def pci_bus_id(dev_index: int):
    """'dddd:bb:dd.f' of the HIP device (lower case), from torch's device properties or hipDeviceGetPCIBusId; None if neither works."""
    try:
        import torch
        props = torch.cuda.get_device_properties(dev_index)
        if hasattr(props, "pci_bus_id") and hasattr(props, "pci_device_id"):
            return f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}."
    except Exception:
        pass
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(dev_index)) == 0:
            return buf.value.decode().lower()[:-1]  # keep 'dddd:bb:dd.'
    except Exception:
        pass
    return None


def sysfs_card(dev_index: int):
    """/sys/class/drm/cardN/device of the HIP device (the host may show more cards than this container's GPU)."""
    import glob
    bus = pci_bus_id(dev_index)
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    for c in cards:
        if bus and bus in os.path.realpath(c).lower() + ".":
            return c
    for c in cards:
        if bus and os.path.basename(os.path.realpath(c)).lower().startswith(bus[:-1]):
            return c
    return None


def bind_to_gpu_numa_node(dev_index: int):
    """Pin this rank to the host cores of its GPU's NUMA node (eight ranks each enqueue ~110 launches per 5 ms step; a rank
    whose launch thread runs on the other socket pays for it in every launch).  The node comes from
    /sys/class/drm/card*/device/numa_node of the PCI device torch reports for `dev_index`; its cores from
    /sys/devices/system/node/nodeN/cpulist.  Guarded: any failure leaves the affinity as it was.  Returns a dict for the line."""
    info = {"numa_node": None, "bound": False, "cpus": None}
    try:
        node = None
        card = sysfs_card(dev_index)
        info["pci"] = pci_bus_id(dev_index)
        if card:
            try:
                node = int(open(os.path.join(card, "numa_node")).read().strip())
            except Exception:
                node = None
        info["numa_node"] = node
        if node is not None and node >= 0 and os.environ.get("NS_BENCH_NO_BIND") != "1":
            cpus = set()
            for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
            allowed = os.sched_getaffinity(0)
            cpus &= allowed
            if cpus and cpus != allowed:
                os.sched_setaffinity(0, cpus)
                info["bound"] = True
        info["cpus"] = len(os.sched_getaffinity(0))
    except Exception as e:  # never fail the run for this
        info["error"] = repr(e)[:120]
    return info


class ClockSampler:
    """Samples the GPU's shader clock / power / busy share from sysfs on a background thread while the sustained leg runs
    (the same box facts `rocm-smi --showclocks` prints; reading sysfs costs the host ~50 us per sample and the GPU nothing)."""

    def __init__(self, dev_index=0, period_s=0.25):
        self.period = period_s
        self.samples = []
        self.dev = sysfs_card(dev_index)
        if self.dev and not os.path.exists(os.path.join(self.dev, "pp_dpm_sclk")):
            self.dev = None
        self._stop = False
        self._thr = None

    def _read(self):
        out = {}
        try:
            for line in open(os.path.join(self.dev, "pp_dpm_sclk")).read().splitlines():
                if line.strip().endswith("*"):
                    out["sclk_mhz"] = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
        except Exception:
            pass
        try:
            out["busy_pct"] = int(open(os.path.join(self.dev, "gpu_busy_percent")).read().strip())
        except Exception:
            pass
        try:
            import glob
            for f in glob.glob(os.path.join(self.dev, "hwmon/hwmon*/power1_average")) + glob.glob(os.path.join(self.dev, "hwmon/hwmon*/power1_input")):
                out["power_w"] = int(open(f).read().strip()) / 1e6
                break
        except Exception:
            pass
        return out

    def __enter__(self):
        if self.dev is None:
            return self
        import threading

        def loop():
            while not self._stop:
                r = self._read()
                if r:
                    self.samples.append(r)
                time.sleep(self.period)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._thr:
            self._thr.join(timeout=2)

    def summary(self):
        def stat(key):
            v = [x[key] for x in self.samples if key in x]
            return None if not v else {"first": v[0], "last": v[-1], "min": min(v), "max": max(v), "mean": round(sum(v) / len(v), 1)}
        return {"source": "sysfs pp_dpm_sclk / gpu_busy_percent / hwmon power of %s, sampled every %.2f s during the leg" % (self.dev, self.period),
                "samples": len(self.samples), "sclk_mhz": stat("sclk_mhz"), "gpu_busy_pct": stat("busy_pct"), "power_w": stat("power_w")}


def run_other_config(name, dev, model=None, sd=None, steps=5, warmup=3, oracle_slice=8, check=True):
    """One BASELINE config other than the headline workload, on the driver's clock: `steps` forwards after `warmup`, the dominant
    kernel (decoder FFN w_1) from the dispatch events of two further forwards, and the oracle beside it on a slice of at most
    `oracle_slice` utterances of the same batch (the full oracle at B = 64 takes minutes): durations / frame counts identical,
    PostNet mel with the bucket decisions pinned.  `model` / `sd`: reuse the headline model when the weights are the same."""
    import smart_nar_fast_tts_amd.workload as wl
    from smart_nar_fast_tts_amd.model import FastSpeech2Align

    cfg_name, B, L, fpp = wl.WORKLOADS[name]
    cfg = wl.model_config(cfg_name)
    t_a = time.perf_counter()
    if model is None:
        sd = wl.synth_state_dict(cfg, seed=0, frames_per_phoneme=fpp) if sd is None else sd
        model = FastSpeech2Align(wl.preprocess_config(), cfg).to(dev).eval()
        model.load_state_dict(sd)
    sp, tx, ln, Lmax = wl.synth_inputs(B, L, seed=0)
    a = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (sp, tx, ln)]
    with torch.no_grad():
        for _ in range(warmup):
            out = model(a[0], a[1], a[2], Lmax)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model(a[0], a[1], a[2], Lmax)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        model.profile_slots((0,))
        for _ in range(2):
            model(a[0], a[1], a[2], Lmax)
        torch.cuda.synchronize()
        k_ms, k_flops, k_n = model.read_profile(0)
        model.profile_slots(())
    frames, T_pad = int(out[9].sum()), int(out[0].shape[1])
    flops_frame = wl.algorithmic_flops_per_frame(cfg, T_pad, L, fpp)
    tf = (k_flops / (k_ms * 1e-3)) / 1e12 if k_ms > 0 else 0.0
    res = {"workload": f"{name}: batch {B}, phoneme_len {L}, T_pad {T_pad}, d_model {cfg['transformer']['decoder_hidden']}, "
                       f"{cfg['transformer']['encoder_layer']}+{cfg['transformer']['decoder_layer']} FFT layers, "
                       f"length regulator {cfg.get('length_regulator', 'hard')}",
           "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 4), "value": round(frames / dt, 1), "unit": "frames/s",
           "T_pad": T_pad, "valid_frames": frames, "rows_phase2": int(model._lib.ns_last_phase2_rows(model._h)),
           "end_to_end_frac_mfma_peak": round(flops_frame * frames / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
           "dominant_kernel": {"kernel": "k_conv_gemm (decoder FFN w_1)", "achieved_tflops": round(tf, 2), "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4),
                               "launches": int(k_n), "avg_launch_ms": round(k_ms / max(k_n, 1), 4),
                               "timed": "HIP events on the dispatch packets of two forwards after the timed steps"}}
    if check:
        from oracle import fs2_oracle as orc

        Bs = min(B, oracle_slice)
        w = orc.to_torch_weights(sd)
        ci = [torch.from_numpy(np.ascontiguousarray(x[:Bs])) for x in (sp, tx, ln)]
        lr = cfg.get("length_regulator", "hard")
        with torch.no_grad():
            ref = orc.forward(w, cfg, ci[0], ci[1], ci[2], Lmax, length_regulator=lr)
            o = model(a[0][:Bs], a[1][:Bs], a[2][:Bs], Lmax)
            pin = model(a[0][:Bs], a[1][:Bs], a[2][:Bs], Lmax, p_targets=ref[2].to(dev), e_targets=ref[3].to(dev))
        valid = ~ref[7].numpy()
        same_shape = tuple(pin[1].shape) == tuple(ref[1].shape)
        diff = (pin[1].cpu() - ref[1]).abs() if same_shape else None
        res["check_vs_oracle"] = {
            "slice": f"the first {Bs} of the {B} utterances, as a batch of their own (both sides)",
            "durations_equal": bool(torch.equal(o[5].cpu(), ref[5])), "frame_counts_equal": bool(torch.equal(o[9].cpu(), ref[9])),
            "valid_frames": int(valid.sum()),
            "postnet_max_abs_buckets_pinned": None if diff is None else float(diff.max()),
            "frames_over_1e-3_buckets_pinned": None if diff is None else int(((diff.amax(dim=2) > 1e-3).numpy() & valid).sum())}
    res["leg_seconds"] = round(time.perf_counter() - t_a, 2)
    return res, model, sd


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
    ap.add_argument("--host-lens", action="store_true", help="hand src_lens to forward() as a HOST tensor (what a caller that collates on the host holds): "
                    "phase 1 of a ragged batch may then run on packed phoneme rows too (traces of that path)")
    ap.add_argument("--batch", type=int, default=0, help="override the workload's per-GPU batch size (sweeps; not a BASELINE config)")
    ap.add_argument("--streams", type=int, default=1, help="issue consecutive steps round-robin on this many HIP streams")
    ap.add_argument("--global-pad", action="store_true", help="pad every shard to the global max mel length (all-reduce MAX)")
    ap.add_argument("--balance", choices=["count", "phonemes"], default=None,
                    help="how the global batch is split over the ranks (sharding.shard_indices): contiguous, or longest-first on the "
                         "phoneme counts; default: contiguous for the uniform BASELINE batch, phonemes with --ragged")
    ap.add_argument("--sustained-s", type=float, default=10.0,
                    help="sustained leg after the timed region: back-to-back forwards for this many seconds or 2000 steps, whichever "
                         "comes first (0 = off; skipped with --no-extras)")
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
    t_init0 = time.perf_counter()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    affinity = bind_to_gpu_numa_node(dev_index) if world > 1 else {"numa_node": None, "bound": False, "cpus": len(os.sched_getaffinity(0))}
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
    # weights: rank 0 packs + uploads (load_state_dict), then ONE broadcast of the arena bytes and an adopt on the other ranks;
    # timed apart so that the first real 8-GPU record says what the "116 MB in one broadcast over xGMI" costs
    t_w0 = time.perf_counter()
    if rank == 0:
        model.load_state_dict(sd)
    torch.cuda.synchronize()
    t_w1 = time.perf_counter()
    arena = model.arena_tensor()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_b0 = time.perf_counter()
    sharding.broadcast_bytes(arena, src=0)
    torch.cuda.synchronize()
    t_b1 = time.perf_counter()
    if rank != 0:
        model.adopt_arena()
    weights_info = {"arena_mb": round(arena.numel() / 1e6, 1), "pack_upload_s_rank0": round(t_w1 - t_w0, 3),
                    "weights_broadcast_ms": round((t_b1 - t_b0) * 1e3, 3) if dist is not None else None}

    # each rank's shard of the global batch (B_shard utterances per GPU): rows [rank*B, (rank+1)*B) of one seeded batch
    ragged = None
    if args.ragged:  # phoneme counts uniform in [L/8, L], one utterance per shard at L: what unbucketed serving batches look like
        rr = np.random.RandomState(7)
        ragged = rr.randint(max(1, L // 8), L + 1, size=B_shard * world)
        ragged[::B_shard] = L
    sp, tx, ln, _ = wl.synth_inputs(B_shard * world, L, seed=0, src_lens=ragged)
    balance = args.balance or ("phonemes" if args.ragged else "count")
    shard_phonemes = [int(np.asarray(ln)[p].sum()) for p in sharding.shard_indices(ln, world, balance)]
    # this rank's share through the sharded entry point's own pieces (sharding.prepare_shard / forward_shard: what
    # sharding.synthesize_sharded composes) — inputs resident on the device before the timed region, and in global-pad mode the
    # deadlock-free exchange (a rank with no utterances, or one whose forward raises, still joins the all-reduce)
    n_all = B_shard * world
    host_batch = ([f"utt{i}" for i in range(n_all)], [""] * n_all, sp, tx, ln, int(tx.shape[1]))
    shard = sharding.prepare_shard(host_batch, dev, balance=balance, host_lens=bool(args.host_lens), world_size=world, rank=rank)
    speakers, texts, src_lens, Lmax = shard.batch[2], shard.batch[3], shard.batch[4], shard.batch[5]
    sp, tx, ln = sp[shard.index], tx[shard.index], ln[shard.index]
    global_pad = bool(args.global_pad and world > 1)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    spin_us = sharding.configure_spin(model, local_world_size=local_world)

    # --streams S > 1: consecutive steps go round-robin onto S HIP streams, so the small-grid phase 1 of step i+1 (and
    # the host read of mel_lens between the phases) overlaps the chip-filling phase 2 of step i.  Same K steps, same work.
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None
    counter = [0]

    def step():
        if streams is None:
            return sharding.forward_shard(model, shard, global_pad)
        st = streams[counter[0] % len(streams)]
        counter[0] += 1
        with torch.cuda.stream(st):
            return sharding.forward_shard(model, shard, global_pad)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t_ready = time.perf_counter()  # device, process group, weights: everything before the first forward
    with torch.no_grad():
        for _ in range(args.warmup):
            out = step()
        fence()
        # the dominant kernel is timed on every PROF_EVERY-th forward of the timed region: a timed launch costs its stream
        # ~5 us (its dispatch packet carries a completion signal with timestamps), 20 us per forward if all four were timed
        PROF_EVERY = 5
        model.profile_slots(())
        # per-step spread without extra syncs: one event per step on the launch stream, read after the closing fence
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if streams is None else None
        t0 = time.perf_counter()
        if marks:
            marks[0].record()
        for i in range(args.steps):
            model.profile_slots((0,) if i % PROF_EVERY == 0 else (), keep=True)
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
                    model(speakers, texts, src_lens, Lmax)
            torch.cuda.synchronize()
            t0p = time.perf_counter()
            for i in range(args.steps):
                with torch.cuda.stream(ps[i % 2]):
                    model(speakers, texts, src_lens, Lmax)
            torch.cuda.synchronize()
            pipelined = time.perf_counter() - t0p
            # ... and in capacity mode (max_mel_len = this batch's padded length, as a server with a bucket ceiling passes it;
            # async_status=True): no host wait inside a forward, so the two streams' forwards overlap wherever their launches
            # fit on the chip together — the tail rounds and the small grids (tools/lab/two_queues.hip).  Same kernels, outputs
            # bit-identical to the synchronous path.
            cap_p = int(out[0].shape[1])
            for i in range(2):
                with torch.cuda.stream(ps[i]):
                    oc_p = model(speakers, texts, src_lens, Lmax, max_mel_len=cap_p, async_status=True)
            torch.cuda.synchronize()
            t0p = time.perf_counter()
            for i in range(args.steps):
                with torch.cuda.stream(ps[i % 2]):
                    oc_p = model(speakers, texts, src_lens, Lmax, max_mel_len=cap_p, async_status=True)
            torch.cuda.synchronize()
            pipelined_cap = (time.perf_counter() - t0p, bool(torch.equal(oc_p[1], out[1]) and oc_p.check() is not None))

    frames = int(out[9].sum().item())  # valid frames of this rank's shard (sum of mel_lens, never B*T_pad)
    T_pad = int(out[0].shape[1])
    stats = torch.tensor([elapsed, float(frames), float(T_pad)], dtype=torch.float64, device=dev)
    devices = [{"rank": rank, "device": f"cuda:{dev_index}", "name": torch.cuda.get_device_name(dev)}]
    # per-rank view (SURVEY.md §8e "scaling risks": per-shard T_pad differs, so load imbalance must be visible in a SCALE line)
    per_rank = [{"rank": rank, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "T_pad": T_pad, "valid_frames": frames,
                 "rows_phase2": int(model._lib.ns_last_phase2_rows(model._h)),  # B*T_pad on the grid, fewer on packed rows (ragged batches)
                 "hsa_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "omp_num_threads": os.environ.get("OMP_NUM_THREADS"),
                 "launcher": os.environ.get("NS_BENCH_LAUNCHER", "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else "none"),
                 "cpu_affinity": affinity, "init_s": round(t_ready - t_init0, 3),
                 # busy-wait budget of the mid-forward hand-over, set from the ranks sharing this host (sharding.spin_budget_us)
                 "spin_us": spin_us, "local_world_size": local_world, "utterances": len(shard)}]
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

    # ---- sustained leg: the same forward back to back for ~10 s (or 2000 steps).  Every headline figure above is a 0.1 s burst;
    # this is what the chip holds once clocks and temperatures have settled.  Per-step times from one event per step on the
    # launch stream; the dominant kernel is timed (dispatch events) during the first and the last 200 steps only.
    sustained_error = None
    def run_sustained():
        # how many steps fill the leg: from 10 calibration steps run now, behind 20 more untimed ones (the K-step burst of a cold process — the driver's first
        # command on a fresh box, `--steps 3` in the contract test — can be twice as slow as the steady state, and a count planned
        # from it ends the leg after half the time asked for); every rank must loop the same count (global-pad steps hold a collective)
        with torch.no_grad():
            for _ in range(20):  # (the clock settles within the first forwards of a process)
                step()
            fence()
            t0c = time.perf_counter()
            for _ in range(10):
                step()
            fence()
        est_t = torch.tensor([min(elapsed_max / args.steps, (time.perf_counter() - t0c) / 10)], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(est_t, op=dist.ReduceOp.MIN)
        est = float(est_t[0])
        n_sus = int(max(50, min(2000, -(-args.sustained_s // est))))
        edge = min(200, n_sus // 4)
        with torch.no_grad(), ClockSampler(dev_index) as clk:
            fence()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_sus + 1)]
            model.profile_dominant_kernel(True)
            t0s = time.perf_counter()
            ev[0].record()
            first = last = None
            for i in range(n_sus):
                if i == edge:
                    first = model.read_profile(0)
                    model.profile_dominant_kernel(False)
                if i == n_sus - edge:
                    model.profile_dominant_kernel(True)
                out_s = step()
                ev[i + 1].record()
            fence()
            sus_elapsed = time.perf_counter() - t0s
            last = model.read_profile(0)
            model.profile_dominant_kernel(False)
        sms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n_sus))
        sstat = torch.tensor([sus_elapsed], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(sstat, op=dist.ReduceOp.MAX)
        sus_max = float(sstat[0])
        quarter = max(1, n_sus // 4)
        order = [ev[i].elapsed_time(ev[i + 1]) for i in range(n_sus)]
        return {"steps": n_sus, "seconds": round(sus_max, 3), "ms_per_step": sus_max / n_sus * 1e3,
                     "value": frames_total * n_sus / sus_max, "unit": "frames/s",
                     "step_ms": {"p50": round(sms[n_sus // 2], 3), "p99": round(sms[min(n_sus - 1, int(n_sus * 0.99))], 3),
                                 "min": round(sms[0], 3), "max": round(sms[-1], 3),
                                 "mean_first_quarter": round(sum(order[:quarter]) / quarter, 3),
                                 "mean_last_quarter": round(sum(order[-quarter:]) / quarter, 3)},
                     "dominant_kernel_avg_us": {"first_steps": edge, "first": round(first[0] / max(first[2], 1) * 1e3, 1) if first else None,
                                                "last_steps": edge, "last": round(last[0] / max(last[2], 1) * 1e3, 1) if last else None},
                     "clock": clk.summary(),
                     "note": "rank 0's shard for the per-step and per-kernel figures; value = whole job over the slowest rank"}


    sustained = None
    if args.sustained_s > 0 and not args.no_extras and streams is None:
        if dist is None:
            try:  # (single process: a failure of this secondary leg must not cost the run its headline line)
                sustained = run_sustained()
            except Exception as e:  # noqa: BLE001
                sustained = None
                sustained_error = repr(e)[:300]
        else:     # (N > 1: the leg contains collectives — an exception on one rank must stop the job, not leave the others waiting)
            sustained = run_sustained()
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
    rocprof_us = None
    tpath = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
    t = cfg["transformer"]
    geom = {"rows": int(out[0].shape[0]) * T_pad, "d_model": t["decoder_hidden"], "d_inner": t["conv_filter_size"],
            "k": t["conv_kernel_size"][0]}
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath))
            if all(rec.get(k) == v for k, v in geom.items()):
                traffic = rec.get("hbm_bytes_per_launch")
                rocprof_us = rec.get("rocprof_timed_avg_us")
        except Exception:
            traffic = None
    res = {
        "metric": "mel_frames_per_sec", "value": value, "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16x3" if b3 else "f32", "data": "synthetic",
        "devices": devices, "backend": backend, "world_size_seen_by_rccl": world_seen, "one_gpu_rig": one_gpu, "per_rank": per_rank,
        "init": dict(weights_info, init_s=round(t_ready - t_init0, 3)),
        "rank_spread": {"ms_per_step_max": max(r["ms_per_step"] for r in per_rank), "ms_per_step_min": min(r["ms_per_step"] for r in per_rank),
                        "max_over_min": round(max(r["ms_per_step"] for r in per_rank) / max(min(r["ms_per_step"] for r in per_rank), 1e-9), 4),
                        # what the split gave every rank to do (sharding.shard_indices): phonemes on the host side, rows of phase 2
                        # (B*T_pad on the grid, the packed windows on ragged batches) and valid frames as measured
                        "balance": balance, "phonemes_per_rank": shard_phonemes,
                        "rows_phase2_max_over_min": round(max(r["rows_phase2"] for r in per_rank) / max(min(r["rows_phase2"] for r in per_rank), 1), 4),
                        "valid_frames_max_over_min": round(max(r["valid_frames"] for r in per_rank) / max(min(r["valid_frames"] for r in per_rank), 1), 4)},
        "config": {"workload": f"{args.workload}{'_bf16x3' if b3 else ''}{' (ragged lengths)' if args.ragged else ''}{f' (batch overridden: {args.batch})' if args.batch > 0 else ''}: LJSpeech config, batch {B_shard}/GPU x {args.gpus} GPU, phoneme_len {L}, "
                               f"T_pad {T_pad_max}, d_model {cfg['transformer']['decoder_hidden']}, "
                               f"{cfg['transformer']['encoder_layer']}+{cfg['transformer']['decoder_layer']} FFT layers, "
                               f"random-init weights (seed 0, duration bias log({fpp + 1:g}))",
                   "global_batch": B_shard * args.gpus, "valid_frames_per_step": int(frames_total),
                   "padding": "global-pad" if global_pad else "per-shard", "shard_balance": balance,
                   "algorithmic_mflop_per_frame": round(flops_frame / 1e6, 2),
                   "end_to_end_tflops": round(flops_frame * value / 1e12, 2)},
        "roofline": {"bound": "mfma", "kernel": "k_conv_gemm (FFN w_1: Conv1d k=9, d->d_inner, bias+ReLU)",
                     "achieved": round(achieved_tflops, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved_tflops / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                     # algorithmic bytes of one launch: activation rows read once, the weight matrix once, the output written once
                     "algorithmic_bytes": (geom["rows"] * geom["d_model"] + geom["d_inner"] * geom["k"] * geom["d_model"] + geom["rows"] * geom["d_inner"]) * 4,
                     "traffic_over_algorithmic": None if traffic is None else round(
                         traffic / ((geom["rows"] * geom["d_model"] + geom["d_inner"] * geom["k"] * geom["d_model"] + geom["rows"] * geom["d_inner"]) * 4), 3),
                     # the same launches in the committed rocprofv3 kernel trace of this command (profiles/, timed region only):
                     # under the profiler they read 1-3 % longer than the in-process events of an unprofiled run
                     "frac_rocprof": None if not rocprof_us or not k_launches else round(
                         (k_flops / k_launches) / (rocprof_us * 1e-6) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                     "rocprof_avg_launch_us": rocprof_us,
                     # where the two figures above that this run did NOT measure come from
                     "traffic_source": None if traffic is None else "committed profile: profiles/dominant_kernel_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh on this launch geometry), replayed, not measured in this run",
                     "frac_rocprof_source": None if not rocprof_us else "committed profile: the same launches' average in the round's rocprofv3 kernel trace (profiles/dominant_kernel_traffic.json rocprof_timed_avg_us), replayed, not measured in this run",
                     "frac_of_measured_peak": round(achieved_tflops / F32_MFMA_MEASURED_TFLOPS, 4),
                     "launches": int(k_launches), "avg_launch_ms": round(k_ms / max(k_launches, 1), 4),
                     "launches_timed": f"every launch of every {PROF_EVERY}th forward of the timed region (HIP events on the dispatch packets)",
                     "share_of_step_time": round((k_ms / len(range(0, args.steps, PROF_EVERY))) / (elapsed / args.steps * 1e3) if elapsed > 0 else 0.0, 3),
                     "geometry": geom},
    }
    if sustained_error:
        res["sustained"] = {"error": sustained_error}
    if sustained:
        res["sustained"] = sustained
        res["burst"] = {"value": value, "ms_per_step": res["ms_per_step"], "steps": args.steps}
        dev_rel = abs(sustained["value"] - value) / value
        sustained["vs_burst"] = round(sustained["value"] / value, 4)
        # the headline stays the K-step figure only while the sustained leg confirms it within 2 %
        if dev_rel > 0.02:
            res["value"] = sustained["value"]
            res["ms_per_step"] = sustained["ms_per_step"]
            res["value_source"] = f"sustained leg ({sustained['steps']} steps): it differs from the {args.steps}-step burst by {100 * dev_rel:.1f} %"
            res["config"]["end_to_end_tflops"] = round(flops_frame * res["value"] / 1e12, 2)
            value = res["value"]
        else:
            res["value_source"] = f"{args.steps}-step timed region (the sustained leg of {sustained['steps']} steps agrees within 2 %)"
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
        res["pipelined"]["capacity_mode"] = {
            "ms_per_step": round(pipelined_cap[0] / args.steps * 1e3, 4), "value": round(frames_total * args.steps / pipelined_cap[0], 1),
            "unit": "frames/s", "bit_identical_to_sync_path": pipelined_cap[1],
            "note": "the same, with forward(max_mel_len=<this batch's padded length>, async_status=True): no host wait inside a forward"}
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
            # independent B=1 requests on several HIP streams of the one model: a B=1 forward is ~80 launches of fewer than 256
            # workgroups, and two queues overlap whenever both launches' workgroups fit on the chip (tools/lab/two_queues.hip)
            conc = {}
            with torch.no_grad():
                for ns in (1, 8):
                    sts = [torch.cuda.Stream(dev) for _ in range(ns)]
                    n_req = 200
                    for i in range(2 * ns + n_req):
                        if i == 2 * ns:
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                        with torch.cuda.stream(sts[i % ns]):
                            ocs = model(a1[0], a1[1], a1[2], L1, max_mel_len=cap, async_status=True)
                    torch.cuda.synchronize()
                    conc[ns] = n_req / (time.perf_counter() - t0)
                    same = bool(torch.equal(ocs[1], o1[1]))
            res["latency_capacity_mode"]["concurrent_streams"] = {
                "utterances_per_s": {str(k): round(v, 1) for k, v in conc.items()}, "streams": 8, "speedup": round(conc[8] / conc[1], 3),
                "bit_identical_to_sync_path": same,
                "note": "200 B=1 forwards round-robin over 1 and 8 streams, one host thread (8: the HIP runtime spreads streams over 4 hardware queues, and 4 streams may share 2 of them — profiles/r04_multi_stream_small.txt); tools/multi_stream_small.py has B=1,2,4 x 1..8 streams"}

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
        rln_host = torch.from_numpy(np.ascontiguousarray(rln_))  # src_lens as a host tensor: phase 1 packs too (nar_fs2.h ns_forward_durations_packed)
        vl = {}
        keep = model.packed_rows
        try:
            # packed: both phases on packed rows (src_lens handed over on the host, as a caller that collates on the host has them);
            # packed_phase2_only: src_lens on the device, as synthesize.py's to_device leaves them (phase 1 stays on the grid);
            # grid: the reference's padded grids throughout
            modes = ("packed", "packed_phase2_only", "grid")

            def run_mode(mode, n):
                model.packed_rows = mode != "grid"
                lens_arg = rln_host if mode == "packed" else ra[2]
                ro = None
                for _ in range(n):
                    ro = model(ra[0], ra[1], lens_arg, L)
                return ro

            with torch.no_grad():
                for mode in modes:  # every mode's scratch, hints and clocks settled before any of them is timed
                    run_mode(mode, 4)
                torch.cuda.synchronize()
                for mode in modes:
                    run_mode(mode, 2)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    ro = run_mode(mode, 10)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / 10
                    vf = int(ro[9].sum())
                    vl[mode] = {"ms_per_step": round(dt * 1e3, 3), "value": round(vf / dt, 1), "unit": "frames/s",
                                "rows_phase2": int(model._lib.ns_last_phase2_rows(model._h)),
                                "phase1_rows": int(model._lib.ns_last_phase1_rows(model._h))}
                    vl["T_pad"], vl["valid_frames"] = int(ro[0].shape[1]), vf
        finally:
            model.packed_rows = keep
        vl["speedup"] = round(vl["grid"]["ms_per_step"] / vl["packed"]["ms_per_step"], 3)
        vl["phase1_rows"] = vl["packed"]["phase1_rows"]
        vl["phase1_rows_grid"] = vl["grid"]["phase1_rows"]
        vl["workload"] = f"{args.workload} with ragged lengths (phoneme counts uniform in [L/8, L], B={B_shard})"
        res["variable_length"] = vl


    if args.gpus == 1 and not args.no_extras and not b3 and args.workload == "cfg2_b16" and args.batch == 0 and not args.ragged:
        # The other BASELINE configs on the driver's clock (config 1 = single utterance, 4 = d_model 512 / 6+6 layers / B = 64,
        # 5 = long-form B = 8 with both length regulators): secondary figures beside the headline workload, each with its own
        # oracle check on a slice.  A failure of one leg is recorded, never fatal to the line.
        others = {}
        check = not args.no_cpu_baseline
        if check:
            torch.set_num_threads(min(os.cpu_count() or 1, 64))
        keep_sd = None
        for oname in ("cfg1_single", "cfg4_d512", "cfg5_longform", "cfg5_longform_gaussian"):
            try:
                if oname == "cfg1_single":     # the headline model's own weights
                    others[oname], _, _ = run_other_config(oname, dev, model=model, sd=sd, check=check)
                elif oname == "cfg5_longform_gaussian":  # config 5's weights, the Gaussian regulator wired in
                    others[oname], _, _ = run_other_config(oname, dev, sd=keep_sd, oracle_slice=2, check=check)
                else:
                    others[oname], om, osd = run_other_config(oname, dev, oracle_slice=2 if oname.startswith("cfg5") else 8, check=check)
                    keep_sd = osd if oname == "cfg5_longform" else None
                    del om
            except Exception as e:  # noqa: BLE001
                others[oname] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
        res["other_configs"] = others

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
