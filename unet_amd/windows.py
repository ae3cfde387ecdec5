"""Window indexing of the reference's datasets and continuous inference -- bit-exact integer math.

    phase_five/sliding_window.py:70,81-82,93-94,201-206   SlidingWindowDataset / process_continuous
    data.py:3103-3115, 3156-3170, 4602-4608                ContinuousLFPDataset / MultiSession / ECoG
    data.py:3275, 4843                                     LOSO temporal split point int(n * 0.7)

plus the sharding of one recording's windows over ranks for multi-GPU sliding-window inference
(SURVEY.md 8e): contiguous window ranges, so each GPU overlap-adds its own span and only the
(window_size - stride)-sample seams need the cross-rank sum.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple


def n_windows_dataset(n_samples: int, window_size: int, stride: int) -> int:
    """data.py:3108 / 3160 / 4604: ``max(0, (n - W) // s + 1)`` (Python floor division: a recording shorter
    than the window gives (negative)//s + 1 <= 0 -> 0)."""
    return max(0, (n_samples - window_size) // stride + 1)


def n_windows_phase5(n_samples: int, window_size: int, stride: int) -> int:
    """phase_five/sliding_window.py:70: ``max(1, ...)`` -- at least one (possibly short) window."""
    return max(1, (n_samples - window_size) // stride + 1)


def n_windows_inference(n_samples: int, window_size: int, stride: int) -> int:
    """phase_five/sliding_window.py:201-206: ``while start + W <= n`` -> 0 when the recording is too short."""
    if n_samples < window_size:
        return 0
    return (n_samples - window_size) // stride + 1


def window_start(idx: int, stride: int) -> int:
    return idx * stride


def coverage(n_samples: int, window_size: int, stride: int) -> float:
    """phase_five/sliding_window.py:90-94."""
    covered = min(n_windows_phase5(n_samples, window_size, stride) * stride + window_size, n_samples)
    return covered / n_samples


def multisession_mapping(session_lengths: Sequence[int], window_size: int, stride: int) -> List[Tuple[int, int]]:
    """data.py:3156-3170: global window index -> (session index, local window index)."""
    out = []
    for si, n in enumerate(session_lengths):
        for w in range(n_windows_dataset(n, window_size, stride)):
            out.append((si, w))
    return out


def ecog_windows(subject_lengths: Sequence[int], window_size: int, stride: int) -> List[Tuple[int, int]]:
    """data.py:4602-4608: (subject index, start sample)."""
    out = []
    for si, n in enumerate(subject_lengths):
        for w in range(n_windows_dataset(n, window_size, stride)):
            out.append((si, w * stride))
    return out


def loso_temporal_split_point(n_samples: int, val_ratio: float = 0.3) -> int:
    """data.py:3275 / 4843: first ``int(n * (1 - val_ratio))`` samples train, the rest validate."""
    return int(n_samples * (1.0 - val_ratio))


@dataclass(frozen=True)
class WindowShard:
    rank: int
    first_window: int
    n_windows: int
    sample_lo: int   # first sample this shard's windows touch
    sample_hi: int   # one past the last


def shard_windows(total_windows: int, world_size: int, window_size: int, stride: int) -> List[WindowShard]:
    """Contiguous, balanced window ranges per rank (earlier ranks take the remainder)."""
    out = []
    base, rem = divmod(total_windows, world_size)
    first = 0
    for r in range(world_size):
        n = base + (1 if r < rem else 0)
        lo = first * stride if n else 0
        hi = (first + n - 1) * stride + window_size if n else 0
        out.append(WindowShard(r, first, n, lo, hi))
        first += n
    return out


class DeviceRecordings:
    """Recordings resident in HBM + batch assembly in one kernel (``unet_gather_windows``).

    The reference's ``ECoGDataset`` (data.py:4574-4612) keeps recordings on the host and ``__getitem__`` slices,
    optionally z-scores and copies ONE window at a time; a DataLoader then stacks and uploads them.  Here every
    recording is uploaded once (channels pre-sliced to the common counts as data.py:4584-4611 does, recordings
    concatenated along time), the window table is the reference's (``ecog_windows``), and ``batch(indices)``
    gathers -- and, if asked, per-window z-scores (numpy's biased std, + 1e-8) -- a whole batch on the device.
    """

    def __init__(self, subjects_data, window_size: int = 5000, stride: int = 2500, zscore_per_window: bool = False,
                 n_source_channels=None, n_target_channels=None, device="cuda", dtype=None):
        import numpy as np
        import torch
        self.window_size, self.stride, self.zscore_per_window = window_size, stride, zscore_per_window
        self.n_source_channels = (min(s["n_source_channels"] for s in subjects_data) if subjects_data else 0) \
            if n_source_channels is None else n_source_channels
        self.n_target_channels = (min(s["n_target_channels"] for s in subjects_data) if subjects_data else 0) \
            if n_target_channels is None else n_target_channels
        lengths = [int(s["n_samples"]) for s in subjects_data]
        self.windows = ecog_windows(lengths, window_size, stride)
        self.offsets = [0]
        for s in subjects_data:
            self.offsets.append(self.offsets[-1] + int(s["source"].shape[1]))
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceRecordings keeps the data in GPU memory; there is no CPU path")
        cat = lambda key, c: torch.from_numpy(np.ascontiguousarray(
            np.concatenate([np.asarray(s[key])[:c] for s in subjects_data], axis=1))) if subjects_data else torch.zeros((c, 0))
        self.source = cat("source", self.n_source_channels).to(self.device)      # ONE upload per recording set
        self.target = cat("target", self.n_target_channels).to(self.device)
        if dtype is not None:
            self.source, self.target = self.source.to(dtype), self.target.to(dtype)
        starts = [self.offsets[si] + st for si, st in self.windows]
        for (si, st), a in zip(self.windows, starts):                              # host-side validation: the kernel trusts these
            if st < 0 or st + window_size > self.offsets[si + 1] - self.offsets[si]:
                raise ValueError(f"window ({si}, {st}) leaves recording {si}")
        self._starts = torch.tensor(starts, dtype=torch.int64, device=self.device)

    def __len__(self) -> int:
        return len(self.windows)

    def _gather(self, rec, idx, out_dtype):
        import torch
        from . import _lib
        from .ops import _dt, _p, _stream
        c, total = rec.shape
        out = torch.empty((idx.numel(), c, self.window_size), dtype=out_dtype or rec.dtype, device=self.device)
        if idx.numel() == 0:
            return out
        starts = self._starts.index_select(0, idx)
        _lib.check(_lib.load().unet_gather_windows(_dt(rec), _dt(out), _p(rec), rec.stride(0), total, _p(starts), idx.numel(), c,
                                                   self.window_size, int(self.zscore_per_window), 1e-8, _p(out), _stream()),
                   "unet_gather_windows")
        return out

    def batch(self, indices, out_dtype=None):
        """{"source": [B, Cs, W], "target": [B, Ct, W], "subject_idx": [B], "start_sample": [B]} on the device."""
        import torch
        idx = torch.as_tensor(indices, dtype=torch.int64)
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= len(self.windows)):
            raise IndexError("window index out of range")
        meta = torch.tensor([self.windows[i] for i in idx.tolist()], dtype=torch.int64).reshape(-1, 2)
        idx = idx.to(self.device)
        return {"source": self._gather(self.source, idx, out_dtype), "target": self._gather(self.target, idx, out_dtype),
                "subject_idx": meta[:, 0], "start_sample": meta[:, 1]}
