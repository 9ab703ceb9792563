"""bench.py's output contract (the driver parses the LAST stdout line as JSON): every required key with the right
type, one short run on the GPU box, CPU-baseline leg included; the plain `python bench.py --gpus N` form launches
its own ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       cwd=ROOT, env=e)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line (rank 0 only)"
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_bench_json_contract():
    d = _run(["--gpus", "1", "--steps", "3", "--warmup", "1"])
    assert d["metric"] == "mel_frames_per_sec" and d["unit"] == "frames/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert isinstance(d["value"], float) and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["workload"].startswith("cfg2_b16") and "model" not in d["config"]
    # value is whole-job valid frames / wall
    assert abs(d["value"] - d["config"]["valid_frames_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["one_gpu_rig"] is False and d["backend"] is None and d["world_size_seen_by_rccl"] == 1
    assert len(d["devices"]) == 1 and d["devices"][0]["device"] == "cuda:0"
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s")
    assert ro["peak"] > 0 and ro["achieved"] > 0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    assert ro["launches"] == 1 * 4  # the dominant kernel runs once per decoder layer; every 5th timed step is timed (steps 0 of 3)
    assert "frac_rocprof" in ro and (ro["frac_rocprof"] is None or 0 < ro["frac_rocprof"] <= 1.0)
    # traffic is the committed PMC figure ONLY when it was measured on this launch geometry
    rec = json.load(open(os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")))
    same = all(rec.get(k) == v for k, v in ro["geometry"].items())
    assert (ro["traffic"] == rec["hbm_bytes_per_launch"]) if same else (ro["traffic"] is None)
    assert ro["algorithmic_bytes"] == (ro["geometry"]["rows"] * 256 + 1024 * 9 * 256 + ro["geometry"]["rows"] * 1024) * 4
    assert (ro["traffic_over_algorithmic"] is None) == (ro["traffic"] is None)
    if ro["traffic"] is not None:
        assert abs(ro["traffic_over_algorithmic"] - ro["traffic"] / ro["algorithmic_bytes"]) < 1e-3
    # both latency figures of config 1: the synchronous forward (the headline p50) and capacity mode beside it
    lat, cap = d["latency"], d["latency_capacity_mode"]
    assert lat["p50_ms"] > 0 and cap["p50_ms"] > 0 and cap["bit_identical_to_sync_path"] is True and cap["status"] == [0]
    pc_ = d["pipelined"]["capacity_mode"]  # two streams in capacity mode: same kernels, same bits, not the headline
    assert pc_["bit_identical_to_sync_path"] is True and pc_["value"] > 0  # (no timing comparison: 3 steps are a cold burst)
    cs = cap["concurrent_streams"]  # independent B=1 requests overlap across streams, and stay bit-identical
    assert cs["bit_identical_to_sync_path"] is True and min(cs["utterances_per_s"].values()) > 0 and set(cs["utterances_per_s"]) == {"1", "8"} and cs["streams"] == 8
    assert [x["rank"] for x in d["per_rank"]] == [0] and d["per_rank"][0]["valid_frames"] == d["config"]["valid_frames_per_step"]
    rb = d["roofline_by_kernel"]
    assert set(rb) == {"ffn_w1", "attention", "postnet_mid"}
    assert rb["attention"]["launches"] == 3 * 4 and rb["postnet_mid"]["launches"] == 3 * 3
    for k in ("attention", "postnet_mid"):
        assert 0 < rb[k]["frac"] <= 1.0 and rb[k]["bound"] == "mfma" and abs(rb[k]["frac"] - rb[k]["achieved"] / rb[k]["peak"]) < 1e-3
    # the three timed kernels cannot add up to more than the step
    assert sum(rb[k]["share_of_step_time"] for k in rb) <= 1.0
    # the sustained leg: >= 10 s or 2000 steps of the same forward, and the headline rule (value stays the K-step figure only
    # while the sustained figure confirms it within 2 %)
    su = d["sustained"]
    assert (su["steps"] >= 2000 or su["seconds"] >= 9.0) and su["value"] > 0 and su["step_ms"]["p50"] <= su["step_ms"]["p99"]
    assert su["dominant_kernel_avg_us"]["first"] > 0 and su["dominant_kernel_avg_us"]["last"] > 0
    assert d["burst"]["steps"] == 3 and d["value_source"]
    if abs(su["value"] - d["burst"]["value"]) / d["burst"]["value"] > 0.02:
        assert d["value"] == su["value"] and "sustained" in d["value_source"]
    else:
        assert d["value"] == d["burst"]["value"]
    assert d["init"]["init_s"] > 0 and d["rank_spread"]["max_over_min"] == 1.0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "frames/s" and cb["sample"]
    chk = d["check_vs_oracle"]
    assert chk["durations_equal"] is True and chk["duration_flips"] == 0 and chk["frame_counts_equal"] is True
    assert chk["postnet_max_abs_buckets_pinned"] < 1e-3 and chk["frames_over_1e-3_buckets_pinned"] == 0
    # the free-running figure is explained by legal bucket flips: none of them away from a bin edge, none by more than one
    assert chk["bucket_flips_off_edge"] == 0 and chk["bucket_flips_by_more_than_one"] == 0
    assert 0 <= chk["bucket_flips"] <= chk["bucket_decisions"] // 100
    if chk["bucket_flips"] == 0:
        assert chk["postnet_max_abs_free_running"] < 1e-3 and chk["frames_over_1e-3_free_running"] == 0
    # the figures this run replays from the committed profile say so on the line
    assert (ro["traffic_source"] is None) == (ro["traffic"] is None) and (ro["traffic"] is None or "committed profile" in ro["traffic_source"])
    assert (ro["frac_rocprof_source"] is None) == (ro["frac_rocprof"] is None)
    # the other BASELINE configs on the same line (configs 1, 4, 5 and config 5 as worded), each timed and checked against the oracle
    oc = d["other_configs"]
    assert set(oc) == {"cfg1_single", "cfg4_d512", "cfg5_longform", "cfg5_longform_gaussian"}
    for name, o in oc.items():
        assert "error" not in o, (name, o)
        assert o["steps"] >= 5 and o["ms_per_step"] > 0 and o["value"] > 0 and o["T_pad"] > 0 and o["valid_frames"] > 0, (name, o)
        assert abs(o["value"] - o["valid_frames"] / (o["ms_per_step"] * 1e-3)) / o["value"] < 1e-3
        assert 0 < o["end_to_end_frac_mfma_peak"] <= 1.0 and 0 < o["dominant_kernel"]["frac"] <= 1.0 and o["dominant_kernel"]["launches"] > 0
        c = o["check_vs_oracle"]
        assert c["durations_equal"] is True and c["frame_counts_equal"] is True and c["frames_over_1e-3_buckets_pinned"] == 0, (name, c)
        assert c["postnet_max_abs_buckets_pinned"] < 1e-3
    assert oc["cfg4_d512"]["workload"].startswith("cfg4_d512: batch 64") and "d_model 512" in oc["cfg4_d512"]["workload"]
    assert oc["cfg5_longform"]["T_pad"] > 3000 and oc["cfg5_longform_gaussian"]["T_pad"] > 3000 and oc["cfg1_single"]["rows_phase2"] < 1100
    rs_ = d["rank_spread"]
    assert rs_["balance"] == "count" and rs_["rows_phase2_max_over_min"] == 1.0 and len(rs_["phonemes_per_rank"]) == 1


@pytest.mark.gpu
def test_bench_other_workload_has_no_stale_traffic():
    """roofline.traffic is a measurement of ONE launch geometry; a different workload must report null, not config 2's number."""
    d = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "cfg1_single", "--no-cpu-baseline", "--sustained-s", "1"])
    assert d["roofline"]["traffic"] is None and d["roofline"]["geometry"]["rows"] != 16160


@pytest.mark.gpu
def test_bench_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the driver's N > 1 form may be either): bench.py re-executes
    itself under torch.distributed.run.  On the one-GPU test box both ranks share cuda:0 over gloo, and the line says so."""
    env = {"NS_BENCH_ONE_GPU": "1"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--sustained-s", "1"], env=env)
    assert d["sustained"]["steps"] >= 50 and d["init"]["weights_broadcast_ms"] is not None  # the sustained leg runs on every rank, fenced
    assert d["n_gpus"] == 2 and d["world_size_seen_by_rccl"] == 2 and d["one_gpu_rig"] is True and d["backend"] == "gloo"
    assert [x["rank"] for x in d["devices"]] == [0, 1] and all(x["device"] == "cuda:0" for x in d["devices"])
    assert d["config"]["global_batch"] == 32 and d["scaling"] == "weak"
    # without the rig switch a 2-rank run on a 1-GPU box must refuse instead of silently sharing a device
    import torch

    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                            "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode != 0 and "no device" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_bench_ragged_shards_are_balanced_by_phonemes():
    """`python bench.py --gpus 2 --ragged` (one-GPU rig): a ragged global batch is split longest-first on the phoneme counts
    (sharding.shard_indices(balance="phonemes"); SURVEY.md section 8e names per-shard imbalance as the scaling risk) and the line
    says so: the balance mode, the phonemes every rank received, and the measured spread of rows and valid frames."""
    d = _run(["--gpus", "2", "--ragged", "--steps", "2", "--warmup", "1", "--no-extras"], env={"NS_BENCH_ONE_GPU": "1"})
    rs = d["rank_spread"]
    assert rs["balance"] == "phonemes" and d["config"]["shard_balance"] == "phonemes" and len(rs["phonemes_per_rank"]) == 2
    assert abs(rs["phonemes_per_rank"][0] - rs["phonemes_per_rank"][1]) <= 8, rs
    assert 1.0 <= rs["valid_frames_max_over_min"] <= 1.05 and rs["rows_phase2_max_over_min"] >= 1.0, rs
    c = _run(["--gpus", "2", "--ragged", "--balance", "count", "--steps", "2", "--warmup", "1", "--no-extras"], env={"NS_BENCH_ONE_GPU": "1"})
    assert c["rank_spread"]["balance"] == "count"
    spread = lambda x: max(x["rank_spread"]["phonemes_per_rank"]) - min(x["rank_spread"]["phonemes_per_rank"])  # noqa: E731
    assert spread(d) <= spread(c)


@pytest.mark.gpu
def test_bench_one_rank_over_rccl():
    """The N > 1 code path over the real backend ("nccl" = RCCL) with the one rank a test box can give it: process-group
    init bound to the device, the weight-arena broadcast, the all-reduces of the timing block and the all-gather of the
    device report, launched the way the driver launches N > 1 (torch.distributed.run, 127.0.0.1)."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    e = dict(os.environ, NS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.pop("NS_BENCH_ONE_GPU", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-extras"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=e)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["world_size_seen_by_rccl"] == 1 and d["one_gpu_rig"] is False
    assert d["devices"] == [{"rank": 0, "device": "cuda:0", "name": d["devices"][0]["name"]}] and d["value"] > 0


@pytest.mark.gpu
def test_bench_torchrun_form_with_8_ranks_sets_its_own_environment():
    """The driver's documented N > 1 form — `python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8` —
    never passes through bench.py's self_launch(), so whatever the ranks need must be set by the ranks themselves before
    torch / HIP / RCCL load.  Run exactly that form with 8 ranks on the one-GPU rig from an environment that LACKS
    HSA_ENABLE_IPC_MODE_LEGACY and OMP_NUM_THREADS and check that every rank reports them, plus its own step time,
    T_pad and frame count (load imbalance across shards must be visible in a SCALE line)."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    e = dict(os.environ, NS_BENCH_ONE_GPU="1")
    for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
              "NS_BENCH_LAUNCHER"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--no-extras"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=e)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["world_size_seen_by_rccl"] == 8 and d["one_gpu_rig"] is True
    pr = d["per_rank"]
    assert [x["rank"] for x in pr] == list(range(8))
    # (torch.distributed.run itself exports OMP_NUM_THREADS=1 for nproc > 1 when the variable is unset; either way every rank has one)
    assert all(x["hsa_ipc_mode_legacy"] == "0" and x["omp_num_threads"] in ("1", "8") and x["launcher"] == "torchrun" for x in pr), pr
    assert all(x["ms_per_step"] > 0 and x["T_pad"] > 900 and x["valid_frames"] > 0 for x in pr)
    assert sum(x["valid_frames"] for x in pr) == d["config"]["valid_frames_per_step"] and d["config"]["global_batch"] == 128
    assert max(x["ms_per_step"] for x in pr) <= d["ms_per_step"] * 1.0001
    # what makes the first real 8-GPU record self-explaining (round-3 review item 7): how long the one broadcast of the packed
    # weights took, how long every rank needed before its first forward, where each rank's host threads may run (the NUMA
    # binding is attempted for N > 1 and must never fail the run), and the spread of the ranks' step times
    init = d["init"]
    assert init["weights_broadcast_ms"] is not None and init["weights_broadcast_ms"] >= 0 and init["arena_mb"] > 100
    assert init["pack_upload_s_rank0"] > 0 and init["init_s"] > 0
    for x in pr:
        aff = x["cpu_affinity"]
        assert set(aff) >= {"numa_node", "bound", "cpus"} and isinstance(aff["bound"], bool) and (aff["cpus"] or 0) >= 1, aff
        assert x["init_s"] > 0
    sp = d["rank_spread"]
    assert sp["ms_per_step_max"] == max(x["ms_per_step"] for x in pr) and sp["ms_per_step_min"] == min(x["ms_per_step"] for x in pr)
    assert sp["max_over_min"] >= 1.0


def test_rank_env_is_set_before_torch_loads():
    """CPU: importing bench.py as a rank of a multi-process job sets the variables RCCL needs on this driver, from the
    ranks' own process (both launch forms), without overriding what the launcher exported."""
    code = ("import os, sys; sys.argv=['bench.py']; import importlib.util as u; "
            "s=u.spec_from_file_location('_b', %r); m=u.module_from_spec(s); "
            "import builtins; real=builtins.__import__\n"
            "def spy(name, *a, **k):\n"
            "    if name == 'torch' and 'seen' not in os.environ: os.environ['seen'] = os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', 'unset') + '/' + os.environ.get('OMP_NUM_THREADS', 'unset')\n"
            "    return real(name, *a, **k)\n"
            "builtins.__import__ = spy; s.loader.exec_module(m); print(os.environ['seen'])") % os.path.join(ROOT, "bench.py")
    base = {k: v for k, v in os.environ.items() if k not in ("HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS", "WORLD_SIZE")}
    for extra, want in (({"WORLD_SIZE": "8"}, "0/8"), ({}, "0/unset"), ({"WORLD_SIZE": "2", "OMP_NUM_THREADS": "3"}, "0/3"),
                        ({"HSA_ENABLE_IPC_MODE_LEGACY": "1"}, "1/unset")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(base, **extra))
        assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == want, (extra, r.stdout, r.stderr[-800:])


def test_self_launch_command_line(monkeypatch):
    """CPU: the re-exec command is the driver's own torchrun form (one node, N ranks, 127.0.0.1 rendezvous)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.self_launch(4) == 7
    c = seen["cmd"]
    assert c[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in c
    assert c[c.index("--master-addr") + 1] == "127.0.0.1" and int(c[c.index("--master-port") + 1]) > 0
    assert c[-4:] == ["--gpus", "4", "--steps", "3"] and c[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
