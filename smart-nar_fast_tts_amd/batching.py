"""The steps either side of the forward: building a batch from phoneme-id arrays and slicing
the padded outputs back into per-utterance results.  Host-side numpy/torch plumbing, restated from

* ``TextDataset.collate_fn``  dataset.py:182-191     -> :func:`collate`
* ``pad_1D``                  utils/tools.py:254-264  -> :func:`pad_1D`
* ``to_device`` (6-tuple)     utils/tools.py:56-63    -> :func:`to_device`
* ``synthesize``              synthesize.py:59-76     -> :func:`synthesize` (forward only: no plots, no vocoder)
* per-utterance slicing       utils/tools.py:153-171  -> :func:`split_outputs`
* ``expand``                  utils/tools.py:100-104  -> :func:`expand`

plus one extension, :func:`bucket_by_length`: batches of similar phoneme length, which cuts padded work and the
batch-composition effects of SURVEY.md F3.
"""
from __future__ import annotations

import numpy as np
import torch


def pad_1D(inputs, PAD: int = 0) -> np.ndarray:
    """Right-pad 1-D arrays with PAD to the longest (utils/tools.py:254-264)."""
    max_len = max(len(x) for x in inputs)
    return np.stack([np.pad(x, (0, max_len - x.shape[0]), mode="constant", constant_values=PAD) for x in inputs])


def collate(data):
    """``data``: list of (basename, speaker_id, phoneme_ids ndarray, raw_text) as TextDataset.__getitem__ yields them
    (dataset.py:157-164).  Returns (ids, raw_texts, speakers, texts, text_lens, max_text_len) (dataset.py:182-191)."""
    ids = [d[0] for d in data]
    speakers = np.array([d[1] for d in data])
    texts = [np.asarray(d[2]) for d in data]
    raw_texts = [d[3] for d in data]
    text_lens = np.array([t.shape[0] for t in texts])
    return ids, raw_texts, speakers, pad_1D(texts), text_lens, max(text_lens)


def to_device(data, device, host_lens: bool = False):
    """The 6-tuple branch of utils/tools.py:56-63: numpy -> torch on ``device``; ids / raw_texts / max_len pass through.
    EXTENSION ``host_lens``: keep ``src_lens`` as a CPU tensor — forward() accepts it, uploads the values itself (no copy command),
    may run phase 1 of a ragged batch on packed phoneme rows, and split_outputs reads the lengths without a device-to-host copy."""
    ids, raw_texts, speakers, texts, src_lens, max_src_len = data
    speakers = torch.from_numpy(np.asarray(speakers)).long().to(device)
    texts = torch.from_numpy(np.asarray(texts)).long().to(device)
    src_lens = torch.from_numpy(np.asarray(src_lens))
    if not host_lens:
        src_lens = src_lens.to(device)
    return ids, raw_texts, speakers, texts, src_lens, max_src_len


def expand(values, durations) -> np.ndarray:
    """Repeat values[i] max(0, int(durations[i])) times (utils/tools.py:100-104)."""
    out = []
    for value, d in zip(values, durations):
        out += [value] * max(0, int(d))
    return np.array(out)


def split_outputs(batch, predictions, preprocess_config):
    """Per-utterance slices of forward()'s 12-tuple, the tensor part of synth_samples (utils/tools.py:153-171):
    mel [mel_len, n_mel] (the reference transposes it for plotting; the time-major slice is returned here),
    duration [src_len], pitch / energy at frame rate (phoneme_level predictions are expanded by the durations)."""
    pp = preprocess_config["preprocessing"]
    src_lens = predictions[8].cpu().tolist()
    mel_lens = predictions[9].cpu().tolist()
    out = []
    for i, basename in enumerate(batch[0]):
        src_len, mel_len = int(src_lens[i]), int(mel_lens[i])
        duration = predictions[5][i, :src_len].detach().cpu().numpy()
        item = {"basename": basename, "mel": predictions[1][i, :mel_len].detach(), "duration": duration,
                "src_len": src_len, "mel_len": mel_len}
        for name, idx in (("pitch", 2), ("energy", 3)):
            if pp[name]["feature"] == "phoneme_level":
                item[name] = expand(predictions[idx][i, :src_len].detach().cpu().numpy(), duration)
            else:
                item[name] = predictions[idx][i, :mel_len].detach().cpu().numpy()
        out.append(item)
    return out


def synthesize(model, batchs, preprocess_config, device="cuda", p_control: float = 1.0, e_control: float = 1.0,
               streams: int = 1, host_lens: bool = False, max_mel_len=None):
    """synthesize.py:59-76 reduced to its tensor contract: to_device -> model(*(batch[2:])) under no_grad ->
    per-utterance results (what synth_samples would plot / vocode).

    EXTENSION: ``streams`` > 1 issues consecutive batches round-robin on that many HIP streams and slices the outputs
    after the last one, so the small-grid phase 1 of batch i+1 and the host read between the phases overlap the
    chip-filling phase 2 of batch i (single utterances on one MI355X: +28 % utterances/s with 2 streams).  Results are
    identical: each forward runs on its own stream with its own scratch.
    EXTENSION: ``host_lens`` keeps ``src_lens`` on the host (see :func:`to_device`); same results to fp32 summation order.
    EXTENSION: ``max_mel_len`` (an int; with ``streams`` > 1) fixes every forward's mel axis (model/modules.py:128-131 ``max_len``
    semantics) so that no forward waits on the host between its phases: the forwards of different streams then overlap
    freely (single utterances: 1.9x utterances/s on 8 streams, profiles/r04_multi_stream_small.txt).  Each batch's result is the
    reference's with ``max_len = max_mel_len``: a batch's longest utterance gains padding, which changes ITS output in the
    reference too (the variance predictors are unmasked between their convolutions) — pass the exact length
    for single utterances whose un-padded result is wanted.  An utterance longer than ``max_mel_len`` raises ValueError once
    the batches are done — it is never silently cut."""
    if max_mel_len is not None and streams <= 1:
        raise ValueError("max_mel_len is the capacity of the multi-stream mode; pass streams > 1 with it")
    if streams <= 1:
        results = []
        for batch in batchs:
            batch = to_device(batch, device, host_lens)
            with torch.no_grad():
                output = model(*(batch[2:]), p_control=p_control, e_control=e_control)
            results.extend(split_outputs(batch, output, preprocess_config))
        return results
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("inputs must live on the MI355X (cuda) device; there is no CPU path")
    pool = [torch.cuda.Stream(device=dev) for _ in range(streams)]
    done = []
    for i, batch in enumerate(batchs):
        with torch.cuda.stream(pool[i % streams]), torch.no_grad():
            batch = to_device(batch, device, host_lens)
            if max_mel_len is None:
                output = model(*(batch[2:]), p_control=p_control, e_control=e_control)
            else:
                output = model(*(batch[2:]), max_mel_len=int(max_mel_len), async_status=True, p_control=p_control, e_control=e_control)
            done.append((batch, output))
    torch.cuda.synchronize(dev)
    results = []
    for batch, output in done:
        if max_mel_len is not None:
            output.check()
        results.extend(split_outputs(batch, output, preprocess_config))
    return results


# A forward's time is (nearly) a function of its ROWS, B * T_pad on the padded grid, through the launch plan (include/nar_fs2.h
# ns_plan_gemm: tiles as tall as the row count asks for, round 5) — not of B.  Whether a group of utterances is cheaper as one batch
# or cut in two is therefore a question to put to the plan's own cost model for the dominant launch (the decoder FFN's k=9
# Conv1D-as-GEMM, ~40 % of a forward), not a list of batch sizes tuned at one utterance length (round 4's (1, 2, 4, 8, 12, 16, 24, 32)
# was right for ~1000-frame utterances only).
# (Both constants are fitted on ONE chip and ONE model: LJSpeech d = 256, 4 decoder layers, MI355X, profiles/r06_batch_size_sweep.txt.
#  Another width or chip keeps the SHAPE of the rule — time follows rows through the plan — and should re-fit them, or pass `cost=`.)
FORWARD_FIXED_US = 450.0   # what a forward costs before its rows count: phase 1 of a small batch, ~55 launch floors, the hand-over
FORWARD_PER_W1 = 2.4       # whole phase 2 over its FFN w_1 launches (config 2: 5.2 ms against 4 x 0.54 ms)
ROWS_LINEAR_US = 0.30      # fallback without the native library: us per phase-2 row (config 2: 5.3 ms over 16 160 rows, less the fixed part)


def forward_cost_us(rows: int, model_config=None, lib=None) -> float:
    """Modelled time of one forward whose phase 2 runs on ``rows`` rows, from the launch plan's estimate for the dominant launch
    (``ns_plan_gemm``'s 8th output, host-side: no GPU needed); below the planner's range (a few hundred rows) the small-grid
    ladder's floor for that launch.  The estimate is the plan's own, including the 1-3 % tie-break margin it prices its candidates
    with.  This is a host-side collate helper: when the native library cannot be loaded (no build on this machine) it falls back to
    a rows-linear cost instead of failing — bucketing then still cuts where padding is removed, only without the plan's steps."""
    import ctypes as C

    from . import _lib, workload as wl

    t = (model_config or wl.LJSPEECH_MODEL_CONFIG)["transformer"]
    d, d_inner, k1, layers = t["decoder_hidden"], t["conv_filter_size"], t["conv_kernel_size"][0], t["decoder_layer"]
    if lib is None:
        try:
            lib = _lib.load()
        except (ImportError, OSError, AttributeError):
            return FORWARD_FIXED_US + ROWS_LINEAR_US * max(int(rows), 0) * (layers / 4.0) * (d / 256.0) ** 2
    o = (C.c_int32 * 8)()
    chunks = k1 * (d // 32)
    if rows > 0 and lib.ns_plan_gemm(int(rows), d_inner, d, k1, o):
        w1 = float(o[7])
    else:  # the K-split ladder: a floor of ~0.6 us per K chunk, then rows at the small tiles' rate
        w1 = max(0.6 * chunks, 0.035 * chunks * rows * (d_inner / 256.0) / 256.0) + 5.0
    return FORWARD_FIXED_US + FORWARD_PER_W1 * layers * w1


def step_friendly_cuts(frames, max_batch: int, cost=forward_cost_us):
    """EXTENSION: cut a group of utterances, given by their (estimated) mel frames in DESCENDING order, into consecutive batches
    of at most ``max_batch`` so that the summed modelled forward time is smallest: batch ``[j, i)`` runs on ``(i - j) * frames[j]``
    rows (padded to its longest member).  Dynamic program over the cut points; returns the batch sizes.  With a launch plan whose
    time follows the rows, a cut only pays when it removes padding worth more than a forward's fixed cost — uniform groups come
    back whole.  Each batch is then one exact forward of the reference's semantics on that batch; which utterances share a batch
    is the caller's choice in the reference too (dataset.py:182-191)."""
    n = len(frames)
    if max_batch < 1:
        raise ValueError("max_batch >= 1")
    if any(frames[i] < frames[i + 1] for i in range(n - 1)):
        raise ValueError("frames must be in descending order (a batch is padded to its first member)")
    best = [0.0] + [float("inf")] * n
    prev = [0] * (n + 1)
    memo = {}
    for i in range(1, n + 1):
        for j in range(max(0, i - max_batch), i):
            rows = (i - j) * int(frames[j])
            if rows not in memo:
                memo[rows] = cost(rows)
            c = best[j] + memo[rows]
            if c < best[i] - 1e-9:
                best[i], prev[i] = c, j
    sizes, i = [], n
    while i > 0:
        sizes.append(i - prev[i])
        i = prev[i]
    return sizes[::-1]


def step_friendly_sizes(n: int, max_batch: int, frames_per_utterance: float = 1000.0, cost=forward_cost_us):
    """``n`` utterances of about ``frames_per_utterance`` mel frames each as batch sizes (see :func:`step_friendly_cuts`)."""
    if n < 0 or max_batch < 1:
        raise ValueError("n >= 0 and max_batch >= 1")
    return step_friendly_cuts([int(frames_per_utterance)] * n, max_batch, cost)


def bucket_by_length(lengths, max_batch: int, max_pad_fraction: float = 0.1, step_friendly: bool = False,
                     frames_per_phoneme: float = 8.0, cost=forward_cost_us):
    """EXTENSION (not in the reference): group utterance indices into batches of similar length.

    Sorted by length, a batch is closed when it holds ``max_batch`` items or when admitting the next item would make
    the shortest member's padding exceed ``max_pad_fraction`` of the batch's max length.  Every index appears once.
    ``step_friendly``: every such group is further cut where the launch plan's cost model says two forwards are cheaper than one
    (:func:`step_friendly_cuts` on ``lengths * frames_per_phoneme`` — the duration predictor has not run yet, phonemes are the
    host-side proxy for frames)."""
    order = np.argsort(np.asarray(lengths), kind="stable")[::-1]
    batches, cur = [], []
    for idx in order:
        if cur:
            longest = lengths[cur[0]]
            if len(cur) >= max_batch or (longest - lengths[idx]) > max_pad_fraction * longest:
                batches.append(cur)
                cur = []
        cur.append(int(idx))
    if cur:
        batches.append(cur)
    if step_friendly:
        cut = []
        for b in batches:
            o = 0
            for s in step_friendly_cuts([int(round(lengths[i] * frames_per_phoneme)) for i in b], max_batch, cost):
                cut.append(b[o:o + s])
                o += s
        batches = cut
    return batches
