"""A minimal FlatBuffers writer for the part of the TFLite schema that a converted Larq model
uses (tensorflow/lite/schema/schema.fbs, file identifier TFL3) -- TEST INFRASTRUCTURE.

There is no flatbuffers library and no .tflite file in the build image, so the reader under
test (compute-engine_amd/csrc/tflite/tflite_flatbuffer_reader.h) is exercised against files
produced here from the same published specification.  Objects are laid out front to back:
a table is preceded by its vtable and followed by the objects it refers to, so every uoffset
is positive as the format requires.  custom_options blobs come from tests/flexbuf.py, which
reproduces the reference's own option bytes (mlir/tests/legalize-lce.mlir:9,21)."""
from __future__ import annotations

import struct

import numpy as np

FLOAT32, INT32, BOOL, INT8 = 0, 2, 6, 9
CUSTOM = 32
_NP2T = {np.dtype(np.float32): FLOAT32, np.dtype(np.int32): INT32, np.dtype(np.bool_): BOOL, np.dtype(np.int8): INT8}


class _Scalar:
    def __init__(self, fmt, value):
        self.fmt, self.value = fmt, value


class _Table:
    def __init__(self, fields):
        self.fields = fields          # {field id: _Scalar | _Table | _Vector | str | None}


class _Vector:
    def __init__(self, kind, items, align=4):
        self.kind, self.items, self.align = kind, items, align   # kind: struct fmt char, "table", or "bytes"


class _Writer:
    def __init__(self):
        self.b = bytearray()

    def pad_to(self, align, offset=0):
        while (len(self.b) + offset) % align:
            self.b.append(0)

    def patch_u32(self, pos, value):
        self.b[pos:pos + 4] = struct.pack("<I", value)

    def write(self, obj) -> int:
        """Writes obj (and everything it refers to); returns the position a uoffset must reach."""
        if isinstance(obj, str):
            raw = obj.encode()
            self.pad_to(4)
            pos = len(self.b)
            self.b += struct.pack("<I", len(raw)) + raw + b"\0"
            return pos
        if isinstance(obj, _Vector):
            return self.write_vector(obj)
        return self.write_table(obj)

    def write_vector(self, v: _Vector) -> int:
        if v.kind == "bytes":
            raw = bytes(v.items)
            self.pad_to(max(4, v.align), offset=4)      # element 0 aligned, length word just before it
            pos = len(self.b)
            self.b += struct.pack("<I", len(raw)) + raw
            return pos
        if v.kind == "table":
            self.pad_to(4)
            pos = len(self.b)
            self.b += struct.pack("<I", len(v.items))
            slots = []
            for _ in v.items:
                slots.append(len(self.b))
                self.b += b"\0\0\0\0"
            for slot, item in zip(slots, v.items):
                target = self.write(item)
                self.patch_u32(slot, target - slot)
            return pos
        size = struct.calcsize("<" + v.kind)
        self.pad_to(max(4, size), offset=4)
        pos = len(self.b)
        self.b += struct.pack("<I", len(v.items))
        for x in v.items:
            self.b += struct.pack("<" + v.kind, x)
        return pos

    def write_table(self, t: _Table) -> int:
        fields = {k: v for k, v in t.fields.items() if v is not None}
        nslots = (max(fields) + 1) if fields else 0
        # table body layout: soffset, then fields in id order
        body = [(None, 4, 4)]                           # (field id, size, align)
        for fid in sorted(fields):
            f = fields[fid]
            size = struct.calcsize("<" + f.fmt) if isinstance(f, _Scalar) else 4
            body.append((fid, size, size))
        offs, cur = {}, 0
        for fid, size, align in body:
            cur = (cur + align - 1) // align * align
            if fid is not None:
                offs[fid] = cur
            cur += size
        tsize = cur
        vsize = 4 + 2 * nslots
        # vtable immediately before the table; the table itself 8-aligned so int64 scalars are
        self.pad_to(2)
        while (len(self.b) + vsize) % 8:
            self.b.append(0)
        vpos = len(self.b)
        self.b += struct.pack("<HH", vsize, tsize)
        for i in range(nslots):
            self.b += struct.pack("<H", offs.get(i, 0))
        tpos = len(self.b)
        self.b += struct.pack("<i", tpos - vpos)
        self.b += b"\0" * (tsize - 4)
        pending = []
        for fid in sorted(fields):
            f, p = fields[fid], tpos + offs[fid]
            if isinstance(f, _Scalar):
                self.b[p:p + struct.calcsize("<" + f.fmt)] = struct.pack("<" + f.fmt, f.value)
            else:
                pending.append((p, f))
        for p, f in pending:
            target = self.write(f)
            self.patch_u32(p, target - p)
        return tpos


class ModelBuilder:
    """builder = ModelBuilder(); t = builder.tensor(...); builder.op(...); data = builder.finish()"""

    def __init__(self, description="lce test model"):
        self.description = description
        self.buffers = [b""]          # buffer 0 is the empty sentinel, as in every .tflite
        self.tensors, self.ops, self.codes = [], [], []
        self.inputs, self.outputs = [], []

    def tensor(self, shape, dtype, name="", data=None, scale=None, zero_point=None) -> int:
        buf = 0
        if data is not None:
            arr = np.ascontiguousarray(data, dtype=dtype)
            assert list(arr.shape) == list(shape)
            self.buffers.append(arr.tobytes())
            buf = len(self.buffers) - 1
        quant = None
        if scale is not None:
            quant = _Table({2: _Vector("f", [float(scale)]), 3: _Vector("q", [int(zero_point or 0)])})
        self.tensors.append(_Table({0: _Vector("i", [int(d) for d in shape]), 1: _Scalar("b", _NP2T[np.dtype(dtype)]),
                                    2: _Scalar("I", buf), 3: name or None, 4: quant}))
        return len(self.tensors) - 1

    def _code(self, custom: str | None, builtin: int) -> int:
        key = (custom, builtin)
        if key not in self.codes:
            self.codes.append(key)
        return self.codes.index(key)

    def custom_op(self, name: str, inputs, outputs, options: bytes) -> int:
        self.ops.append(_Table({0: _Scalar("I", self._code(name, CUSTOM)), 1: _Vector("i", list(inputs)),
                                2: _Vector("i", list(outputs)), 5: _Vector("bytes", options, align=4),
                                6: _Scalar("b", 0)}))
        return len(self.ops) - 1

    def builtin_op(self, builtin_code: int, inputs, outputs) -> int:
        self.ops.append(_Table({0: _Scalar("I", self._code(None, builtin_code)), 1: _Vector("i", list(inputs)),
                                2: _Vector("i", list(outputs))}))
        return len(self.ops) - 1

    def finish(self) -> bytes:
        codes = [_Table({0: _Scalar("b", min(b, 127)), 1: c, 2: _Scalar("i", 1), 3: _Scalar("i", b)})
                 for c, b in self.codes]
        sub = _Table({0: _Vector("table", self.tensors), 1: _Vector("i", self.inputs), 2: _Vector("i", self.outputs),
                      3: _Vector("table", self.ops), 4: "main"})
        bufs = [_Table({0: _Vector("bytes", b, align=16) if b else None}) for b in self.buffers]
        model = _Table({0: _Scalar("I", 3), 1: _Vector("table", codes), 2: _Vector("table", [sub]),
                        3: self.description, 4: _Vector("table", bufs)})
        w = _Writer()
        w.b += b"\0\0\0\0TFL3"
        root = w.write(model)
        w.patch_u32(0, root)
        return bytes(w.b)
