"""Per-hardware-channel option overrides, resolved on the host into per-record SoA arrays.

Semantics follow the reference's layered plugin channel config
(waveform_analysis/core/hardware/channel.py:268-313,379-431):

    channel_config = {                       # optionally wrapped as {run_id: {...}}
        "defaults": {opt: value, ...},       # applies to every channel
        "groups":   [{"name":..., "channels": ["0:1", (0, 2)], "config": {...}}, ...]
                    or {name: {"channels": [...], "config": {...}}},
        "channels": {"0:3": {...}, (0, 4): {...}},   # or the same keys at top level
    }

Precedence: base option values < defaults < matching groups (in order) < the channel entry.
Channel keys are "board:channel" strings or (board, channel) pairs; anything else raises
ValueError("Invalid channel key ...") like the reference (channel.py:229-233).

The kernels never see this structure: plugins call `per_record_option` once per distinct
(board, channel) and scatter the result into a per-record array that is uploaded with the
records SoA.
"""

from __future__ import annotations

from collections.abc import Mapping, Sequence
from typing import Any

import numpy as np

_RESERVED = ("defaults", "groups", "channels")


def parse_channel_key(key: Any):
    """Return (board, channel) or None if `key` is not a channel reference."""
    if isinstance(key, (tuple, list)) and len(key) == 2:
        try:
            return int(key[0]), int(key[1])
        except (TypeError, ValueError):
            return None
    if hasattr(key, "board") and hasattr(key, "channel"):
        return int(key.board), int(key.channel)
    if isinstance(key, str) and ":" in key:
        left, right = key.strip().split(":", 1)
        try:
            return int(left.strip()), int(right.strip())
        except (TypeError, ValueError):
            return None
    return None


def _bad_key(key: Any) -> ValueError:
    return ValueError(
        f"Invalid channel key {key!r}; expected HardwareChannel, (board, channel), "
        'or "board:channel".'
    )


def _groups(block: Mapping) -> list[Mapping]:
    groups = block.get("groups")
    if isinstance(groups, Mapping):
        out = []
        for name, grp in groups.items():
            if isinstance(grp, Mapping):
                out.append(grp if "name" in grp else {"name": str(name), **grp})
        return out
    if isinstance(groups, Sequence) and not isinstance(groups, (str, bytes)):
        return [g for g in groups if isinstance(g, Mapping)]
    return []


def _selects(selectors: Any, hw: tuple[int, int]) -> bool:
    if not isinstance(selectors, Sequence) or isinstance(selectors, (str, bytes)):
        return False
    return any(parse_channel_key(item) == hw for item in selectors)


def resolve_channel_values(
    channel_config: Any,
    run_id: str,
    board: int,
    channel: int,
    base_values: Mapping[str, Any] | None = None,
) -> dict[str, Any]:
    """Effective option values for one hardware channel (channel.py:412-431)."""
    hw = (int(board), int(channel))
    resolved: dict[str, Any] = dict(base_values or {})
    if not isinstance(channel_config, Mapping):
        return resolved
    block = channel_config
    if isinstance(block.get(run_id), Mapping):  # {run_id: {...}} wrapper
        block = block[run_id]

    defaults = block.get("defaults")
    if isinstance(defaults, Mapping):
        resolved.update(defaults)
    for grp in _groups(block):
        if _selects(grp.get("channels"), hw):
            values = grp.get("config")
            if isinstance(values, Mapping):
                resolved.update(values)

    chan_block = block.get("channels")
    if not isinstance(chan_block, Mapping):
        chan_block = block
    for key, values in chan_block.items():
        if isinstance(key, str) and key in _RESERVED:
            continue
        parsed = parse_channel_key(key)
        if parsed is None:
            raise _bad_key(key)
        if parsed != hw:
            continue
        if not isinstance(values, Mapping):
            raise ValueError(
                f"Invalid channel config for {key!r}; expected a mapping, got "
                f"{type(values).__name__}."
            )
        resolved.update(values)
        break
    return resolved


def per_record_option(
    boards: np.ndarray,
    channels: np.ndarray,
    channel_config: Any,
    run_id: str,
    base_values: Mapping[str, Any],
) -> dict[tuple[int, int], dict[str, Any]]:
    """Resolve options once per distinct (board, channel) present in the records."""
    keys = np.stack([np.asarray(boards, dtype=np.int64), np.asarray(channels, dtype=np.int64)], axis=1)
    out: dict[tuple[int, int], dict[str, Any]] = {}
    for b, c in np.unique(keys, axis=0) if len(keys) else ():
        out[(int(b), int(c))] = resolve_channel_values(channel_config, run_id, int(b), int(c), base_values)
    return out


def scatter_per_record(
    boards: np.ndarray,
    channels: np.ndarray,
    per_channel: Mapping[tuple[int, int], Any],
    default: float,
    dtype=np.float64,
) -> np.ndarray:
    """Expand {(board, channel): value} into a per-record array."""
    boards = np.asarray(boards)
    channels = np.asarray(channels)
    out = np.full(len(boards), default, dtype=dtype)
    for (b, c), value in per_channel.items():
        out[(boards == b) & (channels == c)] = value
    return out


__all__ = [
    "parse_channel_key",
    "resolve_channel_values",
    "per_record_option",
    "scatter_per_record",
]
