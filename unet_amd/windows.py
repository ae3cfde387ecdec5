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
