"""Minimal FlexBuffers *writer* for the LCE custom-op option maps (test tooling).

Follows the published FlexBuffers builder algorithm (flexbuffers::Builder::Map / Int /
EndMap / Finish) for a root map of integer values; checked byte-for-byte against the two
option blobs the reference's tests hold (tests/golden/reference_kats.json,
mlir/tests/legalize-lce.mlir:9,21)."""
from __future__ import annotations

FBT_INT, FBT_KEY, FBT_MAP = 1, 4, 9


def _width_u(v: int) -> int:
    for bw in (1, 2, 4, 8):
        if v < (1 << (8 * bw)):
            return bw
    raise ValueError(v)


def _width_i(v: int) -> int:
    for bw in (1, 2, 4, 8):
        if -(1 << (8 * bw - 1)) <= v < (1 << (8 * bw - 1)):
            return bw
    raise ValueError(v)


def _pad(n: int, bw: int) -> int:
    return (-n) % bw


def _offset_width(buf_len: int, target: int, elem_index: int) -> int:
    """Width needed for a relative offset stored as element `elem_index` of a vector that
    starts (after alignment) at the current end of the buffer."""
    for bw in (1, 2, 4, 8):
        loc = buf_len + _pad(buf_len, bw) + elem_index * bw
        if _width_u(loc - target) <= bw:
            return bw
    raise ValueError


def build_int_map(items: dict[str, int], insertion_order: list[str] | None = None) -> bytes:
    buf = bytearray()
    order = insertion_order or list(items)
    key_pos = {}
    for k in order:                      # Builder::Key writes the strings as they arrive
        key_pos[k] = len(buf)
        buf += k.encode() + b"\0"
    keys = sorted(items, key=lambda s: s.encode())   # EndMap sorts by strcmp
    n = len(keys)

    # ---- typed vector of keys (prefix: size) ----
    bw = _width_u(n)
    for i, k in enumerate(keys):
        bw = max(bw, _offset_width(len(buf), key_pos[k], i + 1))
    buf += b"\0" * _pad(len(buf), bw)
    buf += n.to_bytes(bw, "little")
    keys_loc = len(buf)
    for k in keys:
        buf += (len(buf) - key_pos[k]).to_bytes(bw, "little")
    keys_bw = bw

    # ---- values vector (prefix: keys offset, keys byte width, size) ----
    bw = max(_width_u(n), _width_u(keys_bw), _offset_width(len(buf), keys_loc, 0))
    for k in keys:
        bw = max(bw, _width_i(items[k]))
    buf += b"\0" * _pad(len(buf), bw)
    buf += (len(buf) - keys_loc).to_bytes(bw, "little")
    buf += keys_bw.to_bytes(bw, "little")
    buf += n.to_bytes(bw, "little")
    vals_loc = len(buf)
    for k in keys:
        buf += items[k].to_bytes(bw, "little", signed=True)
    for k in keys:                       # one packed type per element: (type << 2) | log2(min width)
        buf.append((FBT_INT << 2) | {1: 0, 2: 1, 4: 2, 8: 3}[_width_i(items[k])])
    vec_bw = bw

    # ---- root ----
    rbw = _offset_width(len(buf), vals_loc, 0)
    buf += b"\0" * _pad(len(buf), rbw)
    buf += (len(buf) - vals_loc).to_bytes(rbw, "little")
    buf.append((FBT_MAP << 2) | {1: 0, 2: 1, 4: 2, 8: 3}[vec_bw])
    buf.append(rbw)
    return bytes(buf)


def bconv2d_options(channels_in, stride_height=1, stride_width=1, dilation_height_factor=1,
                    dilation_width_factor=1, padding=1, pad_values=0, fused_activation_function=0) -> bytes:
    """The attribute map mlir/ir/lce_ops.cc:36-51 writes for LceBconv2d."""
    items = dict(channels_in=channels_in, dilation_height_factor=dilation_height_factor,
                 dilation_width_factor=dilation_width_factor,
                 fused_activation_function=fused_activation_function, pad_values=pad_values,
                 padding=padding, stride_height=stride_height, stride_width=stride_width)
    return build_int_map(items)


def bmaxpool_options(filter_height, filter_width, stride_height, stride_width, padding) -> bytes:
    return build_int_map(dict(padding=padding, stride_width=stride_width, stride_height=stride_height,
                              filter_width=filter_width, filter_height=filter_height))
