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


# Batch sizes at which a forward of ~1000-frame utterances sits at the top of a step of the chip (every large launch fills its last
# round of 256 workgroups): the local minima of ms per utterance in profiles/r04_batch_size_sweep.txt — 4: 0.435, 8: 0.361,
# 12: 0.366, 16: 0.327, 24: 0.330, 32: 0.314 ms — where 9 costs 0.429 and 17 0.379 (what is left of the
# staircase is granularity one summation order per forward cannot buy back; the batch composition CAN avoid it).
STEP_FRIENDLY_SIZES = (1, 2, 4, 8, 12, 16, 24, 32)


def step_friendly_sizes(n: int, max_batch: int, sizes=STEP_FRIENDLY_SIZES):
    """EXTENSION: cut ``n`` utterances into batch sizes from ``sizes`` (largest first, each <= max_batch): 9 -> [8, 1],
    17 -> [16, 1], 20 -> [16, 4], 33 -> [32, 1].  Each batch is then one exact forward of the reference's semantics on that
    batch; which utterances share a batch is the caller's choice in the reference too (dataset.py:182-191)."""
    if n < 0 or max_batch < 1:
        raise ValueError("n >= 0 and max_batch >= 1")
    allowed = sorted({s for s in sizes if 1 <= s <= max_batch} | {1}, reverse=True)
    out = []
    while n > 0:
        s = next(a for a in allowed if a <= n)
        out.append(s)
        n -= s
    return out


def bucket_by_length(lengths, max_batch: int, max_pad_fraction: float = 0.1, step_friendly: bool = False):
    """EXTENSION (not in the reference): group utterance indices into batches of similar length.

    Sorted by length, a batch is closed when it holds ``max_batch`` items or when admitting the next item would make
    the shortest member's padding exceed ``max_pad_fraction`` of the batch's max length.  Every index appears once.
    ``step_friendly``: every such group is further cut into the batch sizes of :func:`step_friendly_sizes`."""
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
            for s in step_friendly_sizes(len(b), max_batch):
                cut.append(b[o:o + s])
                o += s
        batches = cut
    return batches
