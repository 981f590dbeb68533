"""`plyfile` stand-in: the subset the reference uses --

    PlyElement.describe(structured_array, 'vertex')          scene/gaussian_model.py:274, scene/dataset_readers.py:137
    PlyData([el]).write(path)                                scene/gaussian_model.py:275, scene/dataset_readers.py:139
    PlyData.read(path); .elements[0][name]; .elements[0].properties[i].name; plydata['vertex']
                                                             scene/gaussian_model.py:283-330, scene/dataset_readers.py:117-123

File format written (what plyfile emits for a structured array with native little-endian scalars,
[UPSTREAM-FROM-MEMORY]: plyfile is not installed here): the header lines `ply`, `format binary_little_endian 1.0`,
optional `comment ...` / `obj_info ...` lines, `element <name> <count>`, one `property <type> <name>` per field with the
classic PLY type names (`float` for f4, `uchar` for u1, ...), `end_header`, then the records packed without padding.
Reading accepts binary little/big endian and ascii files whose elements have scalar properties only (the splat and
point-cloud files of the reference); list properties raise.
"""
from __future__ import annotations

import numpy as np

# numpy kind+size -> PLY type name, as plyfile writes them
_TO_PLY = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}
_FROM_PLY = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
             "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


class PlyHeaderParseError(Exception):
    pass


class PlyProperty:
    def __init__(self, name: str, val_dtype: str):
        self.name = name
        self.val_dtype = val_dtype   # numpy code without byte order, e.g. 'f4'

    def __repr__(self):
        return f"PlyProperty({self.name!r}, {self.val_dtype!r})"


class PlyElement:
    def __init__(self, name: str, data: np.ndarray, comments=None):
        self.name = name
        self.data = data
        self.comments = list(comments or [])
        self.properties = tuple(PlyProperty(n, data.dtype.fields[n][0].str.lstrip("<>=|")) for n in data.dtype.names)

    @staticmethod
    def describe(data: np.ndarray, name: str, len_types=None, val_types=None, comments=None) -> "PlyElement":
        if not isinstance(data, np.ndarray) or data.dtype.names is None or data.ndim != 1:
            raise TypeError("PlyElement.describe needs a one-dimensional structured numpy array")
        for n in data.dtype.names:
            f = data.dtype.fields[n][0]
            if f.shape != () or f.str.lstrip("<>=|") not in _TO_PLY:
                raise ValueError(f"property {n!r}: only scalar integer / float fields are supported by this plyfile stand-in")
        return PlyElement(name, data, comments)

    @property
    def count(self) -> int:
        return int(self.data.shape[0])

    def __len__(self):
        return self.count

    def __getitem__(self, key):
        return self.data[key]

    def __setitem__(self, key, value):
        self.data[key] = value

    def ply_property(self, name: str) -> PlyProperty:
        for p in self.properties:
            if p.name == name:
                return p
        raise KeyError(name)

    def _header_lines(self):
        lines = [f"element {self.name} {self.count}"] + [f"comment {c}" for c in self.comments]
        return lines + [f"property {_TO_PLY[p.val_dtype]} {p.name}" for p in self.properties]


class PlyData:
    def __init__(self, elements=(), text: bool = False, byte_order: str = "=", comments=None, obj_info=None):
        self.elements = list(elements)
        self.text = text
        self.byte_order = "<" if byte_order in ("=", "<") else ">"
        self.comments = list(comments or [])
        self.obj_info = list(obj_info or [])

    def __getitem__(self, name: str) -> PlyElement:
        for e in self.elements:
            if e.name == name:
                return e
        raise KeyError(name)

    def __contains__(self, name: str) -> bool:
        return any(e.name == name for e in self.elements)

    def __iter__(self):
        return iter(self.elements)

    def __len__(self):
        return len(self.elements)

    # ---- writing ----------------------------------------------------------------------------------
    @property
    def header(self) -> str:
        fmt = "ascii" if self.text else ("binary_little_endian" if self.byte_order == "<" else "binary_big_endian")
        lines = ["ply", f"format {fmt} 1.0"] + [f"comment {c}" for c in self.comments] + [f"obj_info {c}" for c in self.obj_info]
        for e in self.elements:
            lines += e._header_lines()
        return "\n".join(lines + ["end_header"])

    def write(self, stream) -> None:
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "wb") if own else stream
        try:
            f.write((self.header + "\n").encode("ascii"))
            for e in self.elements:
                if self.text:
                    for row in e.data:
                        f.write((" ".join(repr(v.item()) if v.dtype.kind == "f" else str(v.item()) for v in row) + "\n").encode("ascii"))
                else:
                    packed = np.dtype([(p.name, self.byte_order + p.val_dtype) for p in e.properties])
                    f.write(np.ascontiguousarray(e.data.astype(packed, copy=False)).tobytes())
        finally:
            if own:
                f.close()

    # ---- reading ----------------------------------------------------------------------------------
    @staticmethod
    def read(stream) -> "PlyData":
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "rb") if own else stream
        try:
            if f.readline().strip() != b"ply":
                raise PlyHeaderParseError("not a PLY file")
            fmt, comments, obj_info, specs = None, [], [], []
            while True:
                raw = f.readline()
                if not raw:
                    raise PlyHeaderParseError("no end_header")
                tok = raw.decode("ascii").strip().split(None, 1)
                if not tok:
                    continue
                key, rest = tok[0], (tok[1] if len(tok) > 1 else "")
                if key == "end_header":
                    break
                if key == "format":
                    fmt = rest.split()[0]
                elif key == "comment":
                    (specs[-1][2] if specs else comments).append(rest)
                elif key == "obj_info":
                    obj_info.append(rest)
                elif key == "element":
                    name, count = rest.split()
                    specs.append((name, int(count), [], []))
                elif key == "property":
                    parts = rest.split()
                    if parts[0] == "list":
                        raise PlyHeaderParseError("list properties are not supported by this plyfile stand-in")
                    if parts[0] not in _FROM_PLY:
                        raise PlyHeaderParseError(f"unknown property type {parts[0]!r}")
                    specs[-1][3].append((parts[1], _FROM_PLY[parts[0]]))
                else:
                    raise PlyHeaderParseError(f"unexpected header line {raw!r}")
            if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
                raise PlyHeaderParseError(f"unsupported format {fmt!r}")
            order = ">" if fmt == "binary_big_endian" else "<"
            elements = []
            for name, count, el_comments, fields in specs:
                native = np.dtype([(n, "=" + t) for n, t in fields])
                if fmt == "ascii":
                    data = np.empty(count, dtype=native)
                    for r in range(count):
                        vals = f.readline().split()
                        data[r] = tuple(np.dtype(t).type(v) for v, (_, t) in zip(vals, fields))
                else:
                    disk = np.dtype([(n, order + t) for n, t in fields])
                    buf = f.read(disk.itemsize * count)
                    if len(buf) != disk.itemsize * count:
                        raise PlyHeaderParseError(f"element {name}: file ends early")
                    data = np.frombuffer(buf, dtype=disk, count=count).astype(native)
                elements.append(PlyElement(name, data, el_comments))
            return PlyData(elements, text=fmt == "ascii", byte_order=order, comments=comments, obj_info=obj_info)
        finally:
            if own:
                f.close()
