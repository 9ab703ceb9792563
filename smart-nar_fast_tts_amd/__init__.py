"""MI355X-native FastSpeech2 inference forward (drop-in for the reference's
``FastSpeech2Align.forward`` path, model/fastspeech2_align.py:30-100).

Submodules are imported lazily: ``workload`` is pure numpy and importable
anywhere; ``model`` / ``ops`` need the HIP C-ABI library and fail loudly
when it is missing."""
__all__ = ["workload"]
