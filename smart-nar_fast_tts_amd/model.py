"""Drop-in for the reference's ``model.FastSpeech2Align`` on the inference path.

Same constructor arguments, ``forward()`` signature, 12-tuple return and
``state_dict`` key names as ``model/fastspeech2_align.py:13-100`` so that
``synthesize.py:59-76`` (``model(*(batch[2:]))`` under ``torch.no_grad()``) runs
unchanged.  All tensor math happens in the HIP kernels behind the C-ABI
(``include/nar_fs2.h``); PyTorch is the allocator and stream owner only.
"""
from __future__ import annotations

import builtins
import contextlib
import ctypes as C
import functools
import json
import math
import os
import time
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from . import workload as wl


def config_struct(preprocess_config: dict, model_config: dict) -> _lib.NsConfig:
    """The keys ``FastSpeech2Align.__init__`` reads (model/fastspeech2_align.py:16-28,
    model/modules.py:20-77, transformer/Models.py:36-71,176-210) as the C-ABI's ns_config."""
    t = model_config["transformer"]
    vp = model_config["variance_predictor"]
    pp = preprocess_config["preprocessing"]
    for lvl in (pp["pitch"]["feature"], pp["energy"]["feature"]):
        assert lvl in ["phoneme_level", "frame_level"]  # model/modules.py:32-33
    ks = t["conv_kernel_size"]
    return _lib.NsConfig(
        n_vocab=wl.N_SYMBOLS + 1, max_seq_len=model_config["max_seq_len"],
        d_enc=t["encoder_hidden"], n_enc_layer=t["encoder_layer"], n_enc_head=t["encoder_head"],
        d_dec=t["decoder_hidden"], n_dec_layer=t["decoder_layer"], n_dec_head=t["decoder_head"],
        d_inner=t["conv_filter_size"], ffn_k1=ks[0], ffn_k2=ks[1],
        vp_filter=vp["filter_size"], vp_kernel=vp["kernel_size"],
        n_bins=model_config["variance_embedding"]["n_bins"], n_mel=pp["mel"]["n_mel_channels"],
        postnet_dim=wl.POSTNET_DIM, postnet_k=wl.POSTNET_K, postnet_n=wl.POSTNET_N,
        pitch_frame_level=int(pp["pitch"]["feature"] == "frame_level"),
        energy_frame_level=int(pp["energy"]["feature"] == "frame_level"),
        # EXTENSION key (absent from the reference's model.yaml): "gaussian" wires the reference's unused
        # GaussianUpsampling module in place of the hard LengthRegulator
        length_regulator={"hard": 0, "gaussian": 1}[model_config.get("length_regulator", "hard")],
        # EXTENSION key: "bf16x3" opts the large decoder-FFN / PostNet contractions into the split-bf16 matrix-core path
        # (include/nar_fs2.h ns_config.matmul_bf16x3); "fp32" (default) is the reference's arithmetic everywhere
        matmul_bf16x3={"fp32": 0, "bf16x3": 1}[model_config.get("matmul", "fp32")],
        # EXTENSION key (tests): "two_launch" never draws a ticket — LayerNorm / predictor tails / attention merges of small
        # grids run as separate launches instead of last-arriver epilogues (include/nar_fs2.h ns_config.row_epilogue); same bits
        row_epilogue={"fused": 0, "two_launch": 1}[model_config.get("row_epilogue", "fused")],
        # EXTENSION key: when phase 1 packs ragged phoneme rows (host src_lens; include/nar_fs2.h ns_config.phase1_packing)
        phase1_packing={"auto": 0, "always": 1, "never": 2}[model_config.get("phase1_packing", "auto")],
    )


class _OutputBlock:
    """Several output tensors cut from ONE device allocation (a torch.empty costs the host 2-4 us, and the forward's host
    work sits on the latency path of a single utterance): ``ptr(name)`` for the native call, ``view(name)`` for the caller.
    ``layout`` may be called again with smaller shapes (capacity allocation: sizes known only after the hand-over)."""
    _ALIGN = 256

    def __init__(self, specs, device):
        self.layout(specs)
        self.buf = torch.empty(self.nbytes, dtype=torch.uint8, device=device)
        self.base = self.buf.data_ptr()

    def layout(self, specs):
        self.at, self.nbytes = _block_layout(tuple(specs))

    def ptr(self, name) -> C.c_void_p:
        return C.c_void_p(self.base + self.at[name][0]) if name in self.at else C.c_void_p(0)

    def view(self, name) -> torch.Tensor:
        off, n, shape, dtype = self.at[name]
        return self.buf[off:off + n].view(dtype).view(shape)


_ITEMSIZE = {torch.float32: 4, torch.bool: 1, torch.long: 8, torch.int32: 4}


@functools.lru_cache(maxsize=2048)
def _block_layout(specs):
    """((name, shape, dtype), ...) -> ({name: (byte offset, bytes, shape, dtype)}, total bytes); 256-byte aligned entries."""
    off, at = 0, {}
    for name, shape, dtype in specs:
        n = math.prod(shape) * _ITEMSIZE[dtype]
        at[name] = (off, n, shape, dtype)
        off += (n + _OutputBlock._ALIGN - 1) // _OutputBlock._ALIGN * _OutputBlock._ALIGN
    return at, max(off, _OutputBlock._ALIGN)


class ForwardOutput(tuple):
    """The reference's 12-tuple (model/fastspeech2_align.py:87-100) plus, as attributes, what belongs to THIS call only:
    ``status`` — the [B] int32 device tensor of per-utterance NS_STATUS_* words (include/nar_fs2.h) — and ``check()``,
    which raises what a synchronous forward raises on the spot (for forwards issued with ``async_status=True``).  Indexing,
    unpacking and ``len()`` are the plain tuple's, so positional consumers (utils/tools.py:158-171) see no difference."""

    def __new__(cls, items, status=None, n_vocab=0):
        self = super().__new__(cls, items)
        self.status = status
        self._n_vocab = n_vocab
        return self

    def check(self):
        """Synchronises with the forward that produced this output; IndexError for a token id outside the vocabulary
        (nn.Embedding, transformer/Models.py:89), ValueError when an utterance was cut off at ``max_mel_len``.
        Returns the status words as a list."""
        if self.status is None:
            return []
        words = self.status.cpu().tolist()
        bad = [i for i, w in enumerate(words) if w & _lib.STATUS_BAD_TOKEN]
        if bad:
            raise IndexError(f"index out of range in self: token id outside [0, {self._n_vocab}) in utterance(s) {bad}")
        cut = [i for i, w in enumerate(words) if w & _lib.STATUS_TRUNCATED]
        if cut:
            raise ValueError(f"max_mel_len is smaller than the longest utterance: utterance(s) {cut} were cut off")
        return words


class FastSpeech2Align:
    """FastSpeech2 (inference) — HIP/gfx950 implementation of the reference module of the same name.

    Threading: one instance serves ONE host thread at a time (like the reference module under its single-threaded caller,
    synthesize.py:59-76); several HIP streams from that thread are fine.  Use one instance per thread / process otherwise."""

    def __init__(self, preprocess_config: dict, model_config: dict):
        self.model_config = model_config
        self.preprocess_config = preprocess_config
        self._lib = _lib.load()
        self._cfg = config_struct(preprocess_config, model_config)
        h = C.c_void_p()
        _lib.check(self._lib.ns_create(C.byref(self._cfg), C.byref(h)), "ns_create")
        self._h = h
        self._device = None
        self._arena = None
        self._ws = OrderedDict()  # (kind, stream handle) -> scratch tensor, least recently used first
        # phase 2 of a synchronous forward runs on packed rows (variable-length batches: include/nar_fs2.h
        # ns_forward_mel_packed); model_config["padded_rows"] = "dense" or NS_PACKED=0 keeps the reference's padded grid
        self.packed_rows = model_config.get("padded_rows", "packed" if os.environ.get("NS_PACKED", "1") != "0" else "dense") == "packed"
        # EXTENSION key: "separate" returns individually-owned tensors (one clone each) instead of views cut from the forward's two
        # output blocks — for callers that torch.save() single outputs (a view would drag its whole block into the file)
        self.outputs = model_config.get("outputs", "views")
        if self.outputs not in ("views", "separate"):
            raise ValueError("model_config['outputs'] must be 'views' or 'separate'")
        self._t_hint = {}         # (B, L) -> T of the last synchronous forward of that shape (capacity guess for the next one)
        self._ws_need = {}        # (kind, B, L, T) -> bytes (ns_*_ws_bytes is a pure function of the config and these)
        self._sd = OrderedDict()  # host copy of what load_state_dict received (for state_dict() / .to())
        self._user_keys = set()   # inference keys a caller's load_state_dict has supplied so far (strict=True reports the rest)
        self._init_sd = None      # constructor-equivalent random init, drawn ONCE on first need (introspection or first forward);
                                  # kept apart from _sd: it is not something a checkpoint load may count as "already loaded"
        self._loaded = False      # a full inference state dict was accepted (load_state_dict) ...
        self._adopted = False     # ... or the packed arena arrived as bytes (adopt_arena)
        self.training = False
        # VarianceAdaptor.__init__ opens stats.json for the bin edges (model/modules.py:41-71).  They are also
        # state-dict entries, so a checkpoint overrides them; until one is loaded they are only DEFAULTS (kept apart
        # from _sd: two of ~150 keys are not an uploadable state dict).
        self._stats = None
        sp = os.path.join(preprocess_config["path"]["preprocessed_path"], "stats.json")
        if os.path.exists(sp):
            with open(sp) as f:
                self._stats = json.load(f)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.ns_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- nn.Module-shaped surface ------------------------------------------------------------
    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("training is out of scope for this path (SURVEY.md §2); only eval() is supported")
        return self.eval()

    def requires_grad_(self, flag: bool = False):
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("this implementation runs on an MI355X only (device must be 'cuda[:N]'); there is no CPU path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self._device != device:
            old_arena = self._arena
            self._device = device
            self._ws = OrderedDict()
            self._arena = None
            if old_arena is not None or self._loaded:
                # the native handle must never keep pointing at the old device's arena: rebind first (this also marks
                # the handle not-ready), then restore the weights on the new device
                self._bind_arena()
                if self._loaded:
                    self._upload()
                elif self._adopted:  # weights arrived as packed bytes (no host copy): move the bytes, adopt again
                    self._arena.copy_(old_arena)
                    torch.cuda.synchronize(device)
                    self.adopt_arena()
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def state_dict(self):
        """What load_state_dict received — or, before anything was loaded, the constructor-equivalent initialisation that
        parameters() / named_parameters() report and the first forward uploads (an nn.Module's state_dict() on a fresh model
        holds the full set too: torch.save(model.state_dict()) must not write an empty file).  A model whose weights arrived as
        packed arena bytes (adopt_arena) has no per-parameter host copy and raises."""
        return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in self._ensure_host_copy().items())

    # What code that introspects an nn.Module finds (model/fastspeech2_align.py:13-28, utils/model.py:31-35 counts parameters):
    # HOST copies of what load_state_dict received — the device copy is one packed arena (tap-major convolutions, fused QKV,
    # BatchNorm folded into the PostNet), not per-parameter tensors, so writing into these does not change the model; call
    # load_state_dict() for that.  Buffers are what nn.Module keeps out of parameters(): the BatchNorm running statistics.
    @staticmethod
    def _is_buffer(key: str) -> bool:
        return key.endswith((".running_mean", ".running_var", ".num_batches_tracked"))

    def named_parameters(self, prefix: str = "", recurse: bool = True):
        for k, v in self._ensure_host_copy().items():
            if not self._is_buffer(k):
                yield (prefix + ("." if prefix else "") + k, torch.from_numpy(np.array(v)).requires_grad_(False))

    def parameters(self, recurse: bool = True):
        for _, p in self.named_parameters():
            yield p

    def named_buffers(self, prefix: str = "", recurse: bool = True):
        for k, v in self._ensure_host_copy().items():
            if self._is_buffer(k):
                yield (prefix + ("." if prefix else "") + k, torch.from_numpy(np.array(v)))

    def buffers(self, recurse: bool = True):
        for _, b in self.named_buffers():
            yield b

    def _default_init(self):
        """The constructor-equivalent initialisation (model/fastspeech2_align.py:16-28), drawn once: what parameters() reports
        before anything is loaded IS what the first forward uploads (_ensure_weights), and load_state_dict(strict=False) fills
        absent keys from the same draw."""
        if self._init_sd is None:
            self._init_sd = OrderedDict(wl.default_init_state_dict(self.model_config, self._stats))
        return self._init_sd

    def _ensure_host_copy(self):
        """The per-parameter host view for introspection; never touches _sd (what load_state_dict really received)."""
        if self._sd:
            return self._sd
        if self._adopted:
            raise RuntimeError("this model's weights arrived as packed arena bytes (adopt_arena): there is no per-parameter host copy; "
                               "introspect the rank that called load_state_dict()")
        return self._default_init() if self._stats is not None else OrderedDict()

    def modules(self):
        """The native forward is one object: there are no sub-modules to walk (hooks on sub-modules have nothing to attach to)."""
        yield self

    def named_modules(self, memo=None, prefix: str = "", remove_duplicate: bool = True):
        yield prefix, self

    def children(self):
        return iter(())

    def named_children(self):
        return iter(())

    def float(self):
        return self  # float32 is what the path computes in

    def _no_cast(self, what):
        raise NotImplementedError(f"{what}: this path computes in float32 only (the reference's arithmetic, mel max-abs < 1e-3); "
                                  "the opt-in bf16x3 matrix mode is model_config['matmul'] = 'bf16x3', not a dtype cast")

    def half(self):
        self._no_cast("half()")

    def bfloat16(self):
        self._no_cast("bfloat16()")

    def double(self):
        self._no_cast("double()")

    def register_forward_hook(self, *a, **k):
        raise NotImplementedError("forward hooks: wrap forward() instead — the 12-tuple is the only tensor boundary of the native forward")

    register_forward_pre_hook = register_forward_hook

    def _stage(self, key: str, a: np.ndarray):
        """Hand one (already validated) entry to the native staging area."""
        shape = (C.c_int64 * a.ndim)(*a.shape)
        _lib.check(self._lib.ns_set_weight(self._h, key.encode(), C.c_void_p(a.ctypes.data), shape, a.ndim),
                   "load_state_dict")

    def _check(self, key: str, a: np.ndarray) -> str:
        """'' when the native side accepts this key name and shape, else its error text.  No side effect on the model."""
        shape = (C.c_int64 * a.ndim)(*a.shape)
        if self._lib.ns_check_weight(self._h, key.encode(), shape, a.ndim) == 0:
            return ""
        return self._lib.ns_last_error().decode()

    def load_state_dict(self, state_dict, strict: bool = True, *, _internal: bool = False):
        """Accepts the reference's checkpoint["model"] (utils/model.py:21-22).  ``mel_encoder.*`` (training-only
        aligner) and ``num_batches_tracked`` entries are accepted and ignored.  Like ``nn.Module.load_state_dict``:
        a shape mismatch always raises; an unexpected key raises when ``strict`` and is skipped (and returned)
        otherwise; keys no load has supplied so far raise when ``strict`` and keep their current (constructor-equivalent) values
        otherwise — a model that already holds a full caller-supplied state dict accepts partial updates.  Every entry is validated
        BEFORE anything is handed to the native side, so a rejected state dict leaves the loaded weights in use."""
        new = OrderedDict()
        if self._stats is not None and "variance_adaptor.pitch_bins" not in self._user_keys:  # stats.json defaults, until a load supplied bins
            pb, eb = wl.variance_bins(self.model_config, self._stats)
            new["variance_adaptor.pitch_bins"] = pb
            new["variance_adaptor.energy_bins"] = eb
        unexpected, errors = [], []
        for k, v in state_dict.items():
            if k.startswith("mel_encoder.") or k.endswith("num_batches_tracked"):
                continue
            a = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            err = self._check(k, a)
            if err and "unexpected key" in err:
                unexpected.append(k)
                continue
            if err:
                errors.append(err)
                continue
            new[k] = a
        if strict and unexpected:
            errors.append("unexpected key(s): " + ", ".join(repr(k) for k in unexpected))
        if errors:
            raise RuntimeError("load_state_dict: " + "; ".join(errors))
        merged = OrderedDict(self._sd)
        merged.update(new)
        # missing = inference keys no CALLER has ever supplied (this load or an earlier one): the constructor-equivalent draw that
        # _ensure_weights() uploads for a model used before any load, or that parameters() shows, does not count as "loaded" —
        # a truncated checkpoint must say so under strict=True whatever happened to the model before (nn.Module semantics)
        provided = self._user_keys | set(new)
        missing = [k for k in wl.inference_keys(self.model_config) if k not in provided]
        if missing and strict and not _internal:
            raise RuntimeError("load_state_dict: missing key(s): " + ", ".join(missing))
        fill = [k for k in missing if k not in merged]
        if fill:  # strict=False: an nn.Module keeps its constructor's initialisation for keys that did not arrive
            if self._stats is None:
                raise RuntimeError("load_state_dict(strict=False) with missing keys needs <preprocessed_path>/stats.json "
                                   "for the constructor-equivalent initialisation of the rest: " + ", ".join(fill))
            init = self._default_init()
            for k in fill:
                merged[k] = init[k]
        if not _internal:
            self._user_keys = provided
        for k, a in merged.items():
            self._stage(k, a)
        self._sd = merged
        self._loaded, self._adopted = True, False
        if self._device is None and torch.cuda.is_available():
            self._device = torch.device("cuda", torch.cuda.current_device())
        if self._device is not None:
            self._upload(staged=True)
        return missing, unexpected

    def _bind_arena(self):
        nbytes = self._lib.ns_arena_bytes(self._h)
        with torch.cuda.device(self._device):
            self._arena = torch.empty(nbytes, dtype=torch.uint8, device=self._device)
            _lib.check(self._lib.ns_bind_arena(self._h, _lib.ptr(self._arena), nbytes), "ns_bind_arena")

    def _upload(self, staged: bool = False):
        if self._arena is None:
            self._bind_arena()
        with torch.cuda.device(self._device):
            if not staged:
                for k, a in self._sd.items():
                    self._stage(k, a)
            _lib.check(self._lib.ns_finalize_weights(self._h, _lib.stream_ptr(self._device)), "load_state_dict")

    def _ensure_weights(self):
        """The reference constructor leaves a runnable random-init module (model/fastspeech2_align.py:16-28); here the
        equivalent weights are drawn on first use when nothing was loaded (torch's default initialisers under torch's
        global RNG, bins from stats.json)."""
        if self._loaded or self._adopted:
            return
        if self._stats is None:
            raise RuntimeError(
                "weights not loaded: call load_state_dict() (or provide <preprocessed_path>/stats.json for a "
                "random-init model like the reference constructor's, model/modules.py:41-46)")
        self.load_state_dict(self._default_init(), _internal=True)

    # ---- multi-GPU weight replication: rank 0 packs, everyone else adopts the bytes ----
    def arena_tensor(self) -> torch.Tensor:
        if self._arena is None:
            if self._device is None:
                self._device = torch.device("cuda", torch.cuda.current_device())
            self._bind_arena()
        return self._arena

    def adopt_arena(self):
        """The arena bytes arrived from elsewhere (an RCCL broadcast, a file): validate the header and mark the model ready.
        Runs on the model's own device and waits for it first — the bytes may still be in flight on a collective's stream."""
        with (torch.cuda.device(self._device) if self._device is not None else contextlib.nullcontext()):
            if self._device is not None:
                torch.cuda.synchronize(self._device)
            _lib.check(self._lib.ns_adopt_arena(self._h), "ns_adopt_arena")
        if not self._loaded:
            self._adopted = True

    # ---- measurement hook (bench.py roofline leg) ----------------------------------------------------
    def profile_dominant_kernel(self, on: bool = True):
        """Time slot 0 only (the dominant kernel): a timed launch costs its stream ~5 us, so the headline region carries
        four of them per forward, not eleven."""
        self.profile_slots((0,) if on else ())

    def profile_slots(self, slots=(0, 1, 2), keep: bool = False):
        """Time exactly these launch groups (include/nar_fs2.h NS_PROFILE_SLOT); () switches the hook off.  ``keep``: do not
        discard what was recorded so far (sampling: time every n-th forward of a region, read once at its end)."""
        mask = 0x10000 if keep else 0
        for i in slots:
            mask |= 2 << int(i)
        _lib.check(self._lib.ns_profile_enable(self._h, mask), "ns_profile_enable")

    PROFILE_SLOTS = ("ffn_w1", "attention", "postnet_mid")  # include/nar_fs2.h: slots of ns_profile_read_slot

    def read_profile(self, slot: int = 0):
        """(total kernel ms, total algorithmic flops, launches) of one profiled launch group since the last read;
        slot 0 = the FFN k=9 Conv1D-as-GEMM (the dominant kernel), 1 = fused attention, 2 = PostNet 512->512 k=5."""
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        _lib.check(self._lib.ns_profile_read_slot(self._h, int(slot), C.byref(ms), C.byref(fl), C.byref(n)),
                   "ns_profile_read_slot")
        return ms.value, fl.value, n.value

    # ---- forward -------------------------------------------------------------------------------
    MAX_WORKSPACE_STREAMS = 8  # streams whose scratch set is kept alive (enc + dec + length buffers; config 2: ~230 MB per stream)

    def _workspace(self, key: str, nbytes: int, stream_handle=None) -> torch.Tensor:
        # one scratch set per (kind, stream): forwards issued on different streams may run concurrently on the GPU
        # (batching.synthesize pipelines consecutive batches that way) and must not share temporaries.  The cache is
        # LRU-bounded: every synthesize(streams=N) call makes fresh streams, and a dropped set goes back to torch's
        # caching allocator, which only re-issues it in stream order.
        key = (key, stream_handle if stream_handle is not None else torch.cuda.current_stream(self._device).cuda_stream)
        w = self._ws.get(key)
        if w is None or w.numel() < nbytes:
            w = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self._device)
            self._ws[key] = w
        self._ws.move_to_end(key)
        self._evict_streams(self._ws, self.MAX_WORKSPACE_STREAMS)
        return w

    @staticmethod
    def _evict_streams(ws, max_streams: int) -> None:
        """Bound the scratch cache by STREAMS, not entries: while more than ``max_streams`` distinct stream handles hold
        entries, the whole set (enc, dec, pinned and device lengths) of the stream whose most recent use is oldest goes."""
        last_use = {}
        for i, k in enumerate(ws):  # least recently used first
            last_use[k[1]] = i
        for old in sorted(last_use, key=last_use.get)[:max(0, len(last_use) - max_streams)]:
            for k in [k for k in ws if k[1] == old]:
                del ws[k]

    def _pinned_lens(self, B: int, stream_handle=None):
        """[B] int64 in pinned (device-visible) host memory, one buffer per launch stream: phase 1's last kernel writes
        mel_lens there as well, so the forward's single host read is a stream synchronisation.  Returns the tensor and a
        numpy view of the same memory (the host reads max / min through it: no torch dispatch on the hand-over path)."""
        key = ("pin", stream_handle if stream_handle is not None else torch.cuda.current_stream(self._device).cuda_stream)
        t = self._ws.get(key)
        if t is None or t[0].numel() < B:
            buf = torch.empty(max(B, 64), dtype=torch.long, pin_memory=True)
            t = (buf, buf.numpy())
            self._ws[key] = t
        self._ws.move_to_end(key)
        return t[0][:B], t[1][:B]

    def _device_lens(self, B: int, stream_handle) -> torch.Tensor:
        """A device vector [B] int64 per launch stream for src_lens that arrive on the host, filled by ns_upload_lengths (the values
        ride in a kernel's argument block: no copy command — a pinned-staging async copy costs a forward ~35 us, a pageable
        ``.to(device)`` ~80 us)."""
        key = ("lens", stream_handle)
        t = self._ws.get(key)
        if t is None or t.numel() < B:
            t = torch.empty(max(B, 64), dtype=torch.long, device=self._device)
            self._ws[key] = t
        self._ws.move_to_end(key)
        return t[:B]

    def _ws_bytes(self, kind: str, B: int, L: int, T: int) -> int:
        key = (kind, B, L, T)
        n = self._ws_need.get(key)
        if n is None:
            n = (self._lib.ns_encoder_ws_bytes(self._h, B, L) if kind == "enc" else self._lib.ns_decoder_ws_bytes(self._h, B, L, T))
            if len(self._ws_need) > 4096:
                self._ws_need.clear()
            self._ws_need[key] = n
        return n

    def release_workspaces(self):
        """Drop every cached scratch set (they are re-created on demand)."""
        self._ws = OrderedDict()

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    SPIN_US = builtins.float(os.environ.get("NS_SPIN_US", "300"))  # (builtins.: the class defines a float() method above)  # busy-wait budget for the mid-forward hand-over, then block

    _UNWRITTEN = int(np.iinfo(np.int64).min)  # what the pinned mel_lens hold until phase 1's last kernel has written them
    POLL_PINNED = os.environ.get("NS_POLL_PINNED", "1") != "0"  # 0: spin on an event behind the kernel instead (A/B runs)

    def _wait_phase1(self, dev, pin_np=None):
        """Wait until phase 1 has produced ``mel_lens``.  The kernel that computes them also stores them straight into pinned host
        memory (one aligned 8-byte store per utterance), so the host polls THAT memory — pre-filled with a sentinel no length can
        take — instead of an event behind the kernel: the values are visible when the stores land, a few microseconds before the
        kernel's completion signal (end-of-kernel cache write-back + signal + query).  Everything the GPU does next is ordered by
        the stream as before; the host only needs the numbers.  A blocking synchronize sleeps and costs ~50 us of wake-up latency
        per forward (single-utterance p50 1.12 vs 1.07 ms), so the wait spins — but only for SPIN_US microseconds (phase 1 of
        one utterance takes ~0.35 ms): a large batch, or many ranks / server threads sharing the host, must not burn a core and
        hold the GIL for milliseconds.  After the budget the thread blocks in event.synchronize()."""
        if self.SPIN_US > 0:
            deadline = time.perf_counter() + self.SPIN_US * 1e-6
            if pin_np is not None and self.POLL_PINNED:
                while int(pin_np.min()) == self._UNWRITTEN:
                    if time.perf_counter() > deadline:
                        break
                else:
                    return
            else:
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(dev))
                while not done.query():
                    if time.perf_counter() > deadline:
                        break
                else:
                    return
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        done.synchronize()

    def check_status(self, out=None):
        """``out.check()`` for the given forward output; without an argument, for the most recent forward of this model
        (a convenience for single-stream callers: the per-call ``ForwardOutput.status`` is what concurrent callers use)."""
        out = out if out is not None else getattr(self, "_last_out", None)
        return [] if out is None else out.check()

    @property
    def last_status(self):
        out = getattr(self, "_last_out", None)
        return None if out is None else out.status

    def forward(self, speakers, texts, src_lens, max_src_len, mels=None, mel_lens=None, max_mel_len=None,
                p_targets=None, e_targets=None, p_control=1.0, e_control=1.0, *, async_status=False):
        """model/fastspeech2_align.py:30-100, inference branch.  ``speakers`` is accepted and ignored
        (no speaker embedding exists in the reference; multi_speaker is False).  Returns the reference's 12-tuple
        (a ``ForwardOutput``: the tuple plus this call's ``status`` tensor).

        Extensions on ``max_mel_len`` (the reference itself cannot run with it set at inference: its mask is built from
        max(mel_len), model/modules.py:136-137; the `max_len` semantics followed are model/modules.py:128-131,204-213):

        * an ``int``: the mel axis is padded and masked to that length.  SYNCHRONOUS like the default path: ``ValueError`` when
          it is smaller than the longest utterance, ``IndexError`` for a token id outside the vocabulary, both on the spot.
        * a callable ``f(local_max_tensor) -> int`` (e.g. ``sharding.global_max``, multi-GPU global-pad mode):
          synchronous as well, padded to the returned length.
        * an ``int`` together with the keyword-only ``async_status=True``: CAPACITY MODE, an explicit opt-in.  The whole forward
          is enqueued without a host synchronisation (nothing waits for ``mel_lens``).  An utterance longer than the capacity is
          CUT OFF and a bad token id reads embedding row 0; nothing can raise on the spot.  Both are recorded per utterance in
          the returned output's ``status`` tensor: call ``out.check()`` (or ``model.check_status(out)``) before trusting it."""
        if mel_lens is not None:
            raise NotImplementedError(
                "teacher-forced / training branch is out of scope; in the reference it calls an undefined "
                "self._calculate_duration (model/fastspeech2_align.py:57)")
        if not texts.is_cuda:
            raise RuntimeError("inputs must live on the MI355X (cuda) device; there is no CPU path")
        if self._device != texts.device:
            self.to(texts.device)
        self._ensure_weights()
        lib, dev = self._lib, self._device
        B, L = int(texts.shape[0]), int(texts.shape[1])
        if int(max_src_len) != L:
            raise ValueError(f"max_src_len ({int(max_src_len)}) must equal texts.shape[1] ({L})")
        texts_c = texts.long().contiguous()
        # src_lens on the HOST (a CPU tensor, numpy array or list: what a caller that collates on the host holds, dataset.py:182-191)
        # lets phase 1 run on packed phoneme rows for ragged batches; a device tensor is used as it is (reading it would be a sync)
        lens_host = None
        if not (torch.is_tensor(src_lens) and src_lens.is_cuda):
            lens_host = np.ascontiguousarray(src_lens.numpy() if torch.is_tensor(src_lens) else np.asarray(src_lens), dtype=np.int64)
            if lens_host.shape != (B,):
                raise ValueError(f"src_lens must have shape ({B},), got {lens_host.shape}")
            lens_c = None  # uploaded below, through pinned staging on the launch stream (a pageable .to(device) costs ~80 us)
        else:
            lens_c = src_lens.to(device=dev, dtype=torch.long).contiguous()
        # a phoneme_level feature is predicted on the encoder output ([B,L], model/modules.py:117-126),
        # a frame_level one after the length regulator ([B,T], :139-149)
        p_frame, e_frame = bool(self._cfg.pitch_frame_level), bool(self._cfg.energy_frame_level)
        n_mel = self._cfg.n_mel
        f32, u8, i64, i32 = torch.float32, torch.bool, torch.long, torch.int32

        def target(name, t, shape):
            if t is None:
                return None
            if tuple(t.shape) != shape:
                raise ValueError(f"{name} must have shape {shape}, got {tuple(t.shape)}")
            return t.to(device=dev, dtype=torch.float32).contiguous()

        def phase2_outputs(T):
            o = [("mel", (B, T, n_mel), f32), ("post", (B, T, n_mel), f32), ("mel_masks", (B, T), u8)]
            if p_frame:
                o.append(("p_pred", (B, T), f32))
            if e_frame:
                o.append(("e_pred", (B, T), f32))
            return o

        # (the device guard costs the host ~4 us: only taken when the current device is another one)
        with (contextlib.nullcontext() if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)):
            sh = torch.cuda.current_stream(dev).cuda_stream
            st = C.c_void_p(sh)
            if lens_c is None:
                lens_c = self._device_lens(B, sh)
                _lib.check(lib.ns_upload_lengths(C.c_void_p(lens_host.ctypes.data), B, _lib.ptr(lens_c), st), "ns_upload_lengths")
            # Host work is ordered around the GPU's critical path: ONE allocation per phase (the views the caller receives are
            # cut from it after the last launch is enqueued), and everything phase 2 needs that does not depend on T — its
            # output block and scratch at a capacity guessed from the previous forward of this shape — is prepared while
            # phase 1 runs, so that between "mel_lens is readable" and phase 2's first launch there is one numpy max and one
            # ctypes call (tools/host_breakdown.py: 20 us of Python there before, 22 us ahead of the first launch).
            o1 = [("log_d", (B, L), f32), ("d_rounded", (B, L), f32), ("src_masks", (B, L), u8), ("mel_lens", (B,), i64),
                  ("status", (B,), i32)]
            if not p_frame:
                o1.append(("p_pred", (B, L), f32))
            if not e_frame:
                o1.append(("e_pred", (B, L), f32))
            blk1 = _OutputBlock(o1, dev)
            ws_enc = self._workspace("enc", self._ws_bytes("enc", B, L, 0), sh)
            fixed_T = isinstance(max_mel_len, (int, np.integer)) and not isinstance(max_mel_len, bool)
            if async_status and not fixed_T:
                raise ValueError("async_status=True needs max_mel_len=<int>: without a host read the mel axis must be fixed by the caller")
            if fixed_T and int(max_mel_len) < 0:
                raise ValueError(f"max_mel_len ({int(max_mel_len)}) is negative")
            # the pinned host copy of mel_lens: only forwards that READ it on the host hand it to the kernel — a capacity-mode
            # forward still pending on this stream must not write into the words a later synchronous forward is polling
            pin = pin_np = None
            if not (fixed_T and async_status):
                pin, pin_np = self._pinned_lens(B, sh)
                pin_np.fill(self._UNWRITTEN)  # (host write, ahead of the enqueue: the hand-over below polls these words)
            tail1 = (B, L, 1.0, float(p_control), float(e_control),
                     _lib.ptr(None if p_frame else target("p_targets", p_targets, (B, L))),
                     _lib.ptr(None if e_frame else target("e_targets", e_targets, (B, L))),
                     _lib.ptr(ws_enc), ws_enc.numel(), blk1.ptr("log_d"), blk1.ptr("d_rounded"), blk1.ptr("src_masks"),
                     blk1.ptr("mel_lens"), blk1.ptr("p_pred"), blk1.ptr("e_pred"), _lib.ptr(pin), st)
            if lens_host is not None and self.packed_rows:
                _lib.check(lib.ns_forward_durations_packed(self._h, _lib.ptr(texts_c), _lib.ptr(lens_c), C.c_void_p(lens_host.ctypes.data), *tail1),
                           "ns_forward_durations_packed")
            else:
                _lib.check(lib.ns_forward_durations(self._h, _lib.ptr(texts_c), _lib.ptr(lens_c), *tail1), "ns_forward_durations")
            blk2 = ws_dec = None
            lens_on_host = False
            if fixed_T and async_status:
                # CAPACITY MODE (model/modules.py:128-131,204-213 `max_len` semantics): the caller fixes the mel axis, so phase 2
                # is enqueued right behind phase 1 — no event wait, no host read.  What the synchronous path checks on the host
                # (token ids in range, T >= the longest utterance) lands in `status` on the device: check_status() raises later.
                T = int(max_mel_len)
            else:
                # the one device->host read: output shapes depend on max(mel_len)
                # (the reference syncs here too: utils/tools.py:92, plus B*L .item() calls at model/modules.py:222)
                # (a token id outside [0, n_vocab) comes back as mel_len = -1 for its utterance; nn.Embedding raises IndexError)
                # (the kernel that produces mel_lens also wrote them into pinned host memory: a stream sync, no D2H copy)
                hint = int(max_mel_len) if fixed_T else self._t_hint.get((B, L))
                if hint is not None and p_targets is None and e_targets is None:
                    Tc = hint if fixed_T else hint + max(8, hint >> 3)
                    blk2 = _OutputBlock(phase2_outputs(Tc), dev)
                    ws_dec = self._workspace("dec", self._ws_bytes("dec", B, L, Tc), sh)
                self._wait_phase1(dev, pin_np)
                lens_on_host = True
                T = int(pin_np.max())
                if int(pin_np.min()) < 0:
                    bad = [i for i, v in enumerate(pin_np.tolist()) if v < 0]
                    raise IndexError(f"index out of range in self: token id outside [0, {self._cfg.n_vocab}) in utterance(s) {bad}")
                if fixed_T:
                    if int(max_mel_len) < T:
                        cut = [i for i, v in enumerate(pin_np.tolist()) if v > int(max_mel_len)]
                        raise ValueError(f"max_mel_len ({int(max_mel_len)}) is smaller than the longest utterance ({T}): utterance(s) {cut}")
                    T = int(max_mel_len)
                elif callable(max_mel_len):
                    longest = T
                    T = int(max_mel_len(torch.tensor(longest, device=dev)))
                    if T < longest:
                        raise ValueError(f"max_mel_len() returned {T}, smaller than the longest utterance ({longest})")
                else:
                    if len(self._t_hint) > 4096:  # (a server sees many batch shapes: keep the table small)
                        self._t_hint.clear()
                    self._t_hint[(B, L)] = T
                if blk2 is not None and T > Tc:
                    blk2 = ws_dec = None
            if blk2 is None:
                blk2 = _OutputBlock(phase2_outputs(T), dev)
                # (T == 0 is still a call: the native side then only fills `status`)
                ws_dec = self._workspace("dec", self._ws_bytes("dec", B, L, max(T, 1)), sh)
            else:
                blk2.layout(phase2_outputs(T))  # same block, offsets for the actual T (it fits: T <= capacity)
                # the scratch was sized for the capacity guess Tc, and ns_decoder_ws_bytes is NOT monotonic in T (a shorter
                # mel axis can take attention's split-key path, whose partials outweigh everything else): size it for T itself
                need = self._ws_bytes("dec", B, L, max(T, 1))
                if ws_dec.numel() < need:
                    ws_dec = self._workspace("dec", need, sh)
            # forward() hands p_targets / e_targets to the variance adaptor in the inference branch too
            # (model/fastspeech2_align.py:70-78): the embedding then comes from bucketize(target)
            tg0 = target("p_targets", p_targets, (B, T)) if p_frame else None
            tg1 = target("e_targets", e_targets, (B, T)) if e_frame else None
            tail = (float(p_control), float(e_control), _lib.ptr(tg0), _lib.ptr(tg1), _lib.ptr(ws_enc), _lib.ptr(ws_dec), ws_dec.numel(),
                    blk2.ptr("mel"), blk2.ptr("post"), blk2.ptr("p_pred") if p_frame else None,
                    blk2.ptr("e_pred") if e_frame else None, blk2.ptr("mel_masks"), blk1.ptr("status"), st)
            if lens_on_host and self.packed_rows:
                # variable-length batches: phase 2 on packed rows (include/nar_fs2.h ns_forward_mel_packed) — the native side
                # takes the row count from the host copy of mel_lens and falls back to the dense grid when packing saves < 10 %
                _lib.check(lib.ns_forward_mel_packed(self._h, B, L, T, blk1.ptr("mel_lens"), _lib.ptr(pin), *tail), "ns_forward_mel_packed")
            else:
                _lib.check(lib.ns_forward_mel(self._h, B, L, T, blk1.ptr("mel_lens"), *tail), "ns_forward_mel")
            # the GPU is busy with phase 2 from here on: cut the caller's tensors out of the two blocks
            log_d, d_rounded, src_masks = blk1.view("log_d"), blk1.view("d_rounded"), blk1.view("src_masks")
            out_mel_lens, status = blk1.view("mel_lens"), blk1.view("status")
            mel, post, mel_masks = blk2.view("mel"), blk2.view("post"), blk2.view("mel_masks")
            p_pred = blk2.view("p_pred") if p_frame else blk1.view("p_pred")
            e_pred = blk2.view("e_pred") if e_frame else blk1.view("e_pred")
        items = (mel, post, p_pred, e_pred, log_d, d_rounded, src_masks, mel_masks, src_lens, out_mel_lens, None, None)
        if self.outputs == "separate":  # own storage per tensor (src_lens is the caller's own tensor, passed through like the reference does)
            with torch.cuda.stream(torch.cuda.current_stream(dev)):
                items = tuple(t.clone() if (torch.is_tensor(t) and i != 8) else t for i, t in enumerate(items))
                status = status.clone()
        out = ForwardOutput(items, status=status, n_vocab=self._cfg.n_vocab)
        # check_status() without an argument needs the status words only: keep those, not the output blocks (a long-form batch's
        # mel / PostNet tensors would otherwise stay pinned until the next forward after the caller has dropped them)
        self._last_out = ForwardOutput((), status=status, n_vocab=self._cfg.n_vocab)
        return out
