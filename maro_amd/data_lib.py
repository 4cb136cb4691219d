"""Native reader / writer of MARO's binary data files (``*.bin`` made by ``maro data build`` / ``BinaryConverter``), so that
CIM dump / real-data folders and citi_bike build folders can be compiled for the engines WITHOUT a MARO checkout.

On-disk format (reference: ``maro/data_lib/common.py:14-32``, ``binary_reader.py:218-295, 327-346``, ``item_meta.py:112-145,
220-226``), little endian:

    header   struct "<4s b I Q I QQ QQ qq" (69 bytes): b"MARO", file_type (1 = single), version (100), item_count, item_size,
             meta_offset, meta_size, data_offset, data_size, starttime, endtime (UTC seconds)
    meta     YAML: ``attributes: [!MaroAttribute {name, dtype, raw_name, slot, adjust_ratio, tzone}, ...]``, ``events``, ...
             dtype codes i / i4 -> int32, i2 -> int16, i8 -> int64, f -> float32, d -> float64 (``dtype_pack_map``)
    data     item_count records of the packed attributes, in file order (readers filter / pick by the ``timestamp`` attribute)

``read_binary`` returns the records as ONE numpy structured array (zero-copy view of the file bytes): the batch engines
want whole columns, not an item iterator.  ``pick_ticks`` reproduces ``ItemTickPicker`` (``binary_reader.py:80-112``) as a
vectorised tick assignment.  ``write_binary`` produces files the reference's ``BinaryReader`` accepts (round-trip tests and
synthetic topologies).
"""
from __future__ import annotations

import struct
from typing import Dict, Optional, Tuple

import numpy as np
import yaml

HEADER = struct.Struct("<4s b I Q I QQ QQ qq")
VERSION = 100
_NP = {"i": "<i4", "i4": "<i4", "i2": "<i2", "i8": "<i8", "f": "<f4", "d": "<f8"}
_UNIT = {"s": 1, "m": 60, "h": 3600, "d": 86400}


class _Loader(yaml.SafeLoader):
    """SafeLoader that turns the reference's ``!MaroAttribute`` / ``!MaroEvent`` tagged mappings into plain dicts."""


_Loader.add_multi_constructor("!Maro", lambda loader, suffix, node: loader.construct_mapping(node, deep=True))
for _tag in ("!MaroAttribute", "!MaroEvent"):   # explicit too: they win over constructors a MARO import registers on yaml.SafeLoader
    _Loader.add_constructor(_tag, lambda loader, node: loader.construct_mapping(node, deep=True))


def read_binary(path: str) -> Tuple[dict, np.ndarray]:
    """(header dict, records): records is a structured array with one field per attribute of the file's meta."""
    with open(path, "rb") as fp:
        raw = fp.read()
    if len(raw) < HEADER.size:
        raise ValueError(f"{path}: shorter than a MARO binary header")
    name, ftype, version, n, item_size, meta_off, meta_size, data_off, data_size, t0, t1 = HEADER.unpack_from(raw)
    if name != b"MARO":
        raise ValueError(f"{path}: not a MARO binary file (magic {name!r})")
    if ftype != 1:
        raise ValueError(f"{path}: only single-file binaries are supported (file_type {ftype})")
    meta = yaml.load(raw[meta_off:meta_off + meta_size].decode(), Loader=_Loader) or {}
    attrs = meta.get("attributes") or []
    dt = np.dtype([(a["name"], _NP[a["dtype"]]) for a in attrs])
    if dt.itemsize != item_size:
        raise ValueError(f"{path}: item size {item_size} does not match its meta ({dt.itemsize})")
    n_have = min(n, data_size // item_size if item_size else 0, max(0, (len(raw) - data_off)) // item_size if item_size else 0)
    rec = np.frombuffer(raw, dtype=dt, count=n_have, offset=data_off)
    hdr = dict(version=version, item_count=n, item_size=item_size, starttime=t0, endtime=t1, attributes=attrs,
               events=meta.get("events") or [], raw_names={a["name"]: a.get("raw_name") for a in attrs})
    return hdr, rec


def write_binary(path: str, columns: Dict[str, np.ndarray], dtypes: Dict[str, str], starttime: Optional[int] = None,
                 endtime: Optional[int] = None, raw_names: Optional[Dict[str, str]] = None) -> None:
    """Write a single-file MARO binary: `columns` name -> array (same length, in this order; one must be ``timestamp``),
    `dtypes` name -> dtype code ("i", "i2", "i8", "f", "d")."""
    names = list(columns)
    n = len(columns[names[0]])
    dt = np.dtype([(k, _NP[dtypes[k]]) for k in names])
    rec = np.zeros(n, dtype=dt)
    for k in names:
        rec[k] = columns[k]
    ts = np.asarray(columns["timestamp"], np.int64)
    t0 = int(ts.min()) if starttime is None and n else int(starttime or 0)
    t1 = int(ts.max()) if endtime is None and n else int(endtime or 0)
    attrs = "".join(f"- !MaroAttribute\n  adjust_ratio: null\n  dtype: {dtypes[k]}\n  name: {k}\n  raw_name: {(raw_names or {}).get(k, k)}\n"
                    f"  slot: 1\n  tzone: null\n" for k in names)
    meta = ("attributes:\n" + attrs + "default_event_name: null\nevent_attr_name: null\nevents: []\n").encode()
    data = rec.tobytes()
    meta_off = HEADER.size
    data_off = meta_off + len(meta)
    with open(path, "wb") as fp:
        fp.write(HEADER.pack(b"MARO", 1, VERSION, n, dt.itemsize, meta_off, len(meta), data_off, len(data), t0, t1))
        fp.write(meta)
        fp.write(data)


def pick_ticks(timestamp: np.ndarray, starttime: int, n_ticks: int, time_unit: str = "m") -> np.ndarray:
    """``ItemTickPicker`` (binary_reader.py:80-112) for a whole column: the tick each record is yielded at when ticks
    0 .. n_ticks-1 are picked in order, or -1 if it is never yielded.  The picker walks the file once; a record is yielded at
    tick t iff starttime + t*unit <= timestamp < starttime + (t+1)*unit; a record older than the tick being picked is
    silently dropped, and one that lies ahead blocks everything behind it until its tick comes."""
    unit = _UNIT[time_unit]
    ts = np.asarray(timestamp, np.int64)
    tick = (ts - starttime) // unit
    out = np.full(ts.shape, -1, np.int64)
    if ts.size == 0 or bool(np.all(tick[1:] >= tick[:-1])):   # time-sorted file (the converter sorts): no record is dropped or blocked
        ok = (tick >= 0) & (tick < n_ticks)
        out[ok] = tick[ok]
        return out
    cur = 0   # the tick the picker is at when it reaches record i
    for i, t in enumerate(tick.tolist()):   # (one pass; the files are sorted in practice, so cur just follows tick)
        if t < cur:
            continue          # "here we can log items that not sorted": dropped
        if t >= n_ticks:
            break             # cached forever: nothing behind it is ever reached
        cur = t
        out[i] = t
    return out


def items_in_range(rec: np.ndarray, starttime: int, start_offset: int = 0, end_offset: Optional[int] = None, endtime: Optional[int] = None,
                   time_unit: str = "s") -> np.ndarray:
    """``BinaryReader.items(start, end, unit)`` (binary_reader.py:218-295): records with start <= timestamp <= end, where the
    scan stops at the first record beyond `end` (the files are time sorted)."""
    unit = _UNIT[time_unit]
    lo = starttime + start_offset * unit
    hi = endtime if end_offset is None else starttime + end_offset * unit
    ts = rec["timestamp"].astype(np.int64)
    beyond = np.flatnonzero(ts > hi)
    stop = int(beyond[0]) if beyond.size else len(ts)
    keep = (ts[:stop] >= lo)
    return rec[:stop][keep]
