"""On-disk formats either side of the hot path: the reference's splat PLY
(scene/gaussian_model.py:236-275 writer, :282-332 reader -- binary little-endian, one `vertex` element, all
properties float32, `f_rest` channel-major, `binding_0` stored as float and read back as int32) and the
`flame_param.npz` next to it (scene/flame_gaussian_model.py:61-71,219-237).

The reference goes through `plyfile` (not installed here) with a per-property Python loop; this reader maps the
file as one structured numpy array, so a 100k-splat avatar loads in milliseconds.  Files written here load in the
reference and vice versa.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np


def ply_property_names(n_rest: int, bound: bool):
    """Property order of the reference's construct_list_of_attributes (:236-251)."""
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    if bound:
        names.append("binding_0")
    return names


def save_ply(path: str, arrs: Dict[str, np.ndarray]) -> None:
    """arrs: the leaf tensors by their reference names (_xyz (N,3), _features_dc (N,1,3), _features_rest (N,K,3),
    _opacity (N,1), _scaling (N,3), _rotation (N,4), optional binding (N,))."""
    xyz = np.asarray(arrs["_xyz"], np.float32)
    N = xyz.shape[0]
    f_dc = np.asarray(arrs["_features_dc"], np.float32).transpose(0, 2, 1).reshape(N, -1)        # (N,3,1) flattened
    f_rest = np.asarray(arrs["_features_rest"], np.float32).transpose(0, 2, 1).reshape(N, -1)    # channel-major (N,3,K)
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, np.asarray(arrs["_opacity"], np.float32).reshape(N, 1),
            np.asarray(arrs["_scaling"], np.float32), np.asarray(arrs["_rotation"], np.float32)]
    binding = arrs.get("binding")
    if binding is not None:
        cols.append(np.asarray(binding).astype(np.float32).reshape(N, 1))
    table = np.ascontiguousarray(np.concatenate(cols, axis=1).astype("<f4"))
    names = ply_property_names(f_rest.shape[1], binding is not None)
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {N}\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1", "char": "i1", "int8": "i1",
              "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def read_ply_table(path: str) -> np.ndarray:
    """The `vertex` element of a binary little-endian PLY as a structured array (zero-copy memory map)."""
    with open(path, "rb") as f:
        head = b""
        while not head.endswith(b"end_header\n"):
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: no end_header")
            head += line
        offset = f.tell()
    lines = head.decode("ascii").splitlines()
    if lines[0].strip() != "ply" or "binary_little_endian" not in lines[1]:
        raise ValueError(f"{path}: only binary_little_endian PLY is supported (what the reference writes)")
    n, fields, in_vertex = None, [], False
    for ln in lines[2:]:
        tok = ln.split()
        if not tok:
            continue
        if tok[0] == "element":
            in_vertex = tok[1] == "vertex"
            if in_vertex:
                n = int(tok[2])
            elif n is not None:
                break   # the vertex element comes first; later elements are ignored
        elif tok[0] == "property" and in_vertex:
            if tok[1] == "list":
                raise ValueError("list properties are not part of the splat format")
            fields.append((tok[2], _PLY_TYPES[tok[1]]))
    if n is None:
        raise ValueError(f"{path}: no vertex element")
    return np.memmap(path, dtype=np.dtype(fields), mode="r", offset=offset, shape=(n,))


def load_ply(path: str, sh_degree: Optional[int] = None) -> Dict[str, np.ndarray]:
    """-> dict with the reference's leaf names/shapes (what GaussianModel.load_arrays takes); `binding` is int32."""
    t = read_ply_table(path)
    names = t.dtype.names
    col = lambda n: np.asarray(t[n], np.float32)
    N = t.shape[0]
    rest = sorted((n for n in names if n.startswith("f_rest_")), key=lambda s: int(s.split("_")[-1]))
    if sh_degree is not None and len(rest) != 3 * (sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {len(rest)} f_rest properties do not match sh_degree {sh_degree}")
    K = len(rest) // 3
    f_rest = np.stack([col(n) for n in rest], 1).reshape(N, 3, K) if K else np.zeros((N, 3, 0), np.float32)
    out = dict(
        _xyz=np.stack([col("x"), col("y"), col("z")], 1),
        _features_dc=np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], 1)[:, None, :],
        _features_rest=np.ascontiguousarray(f_rest.transpose(0, 2, 1)),
        _opacity=col("opacity")[:, None],
        _scaling=np.stack([col(n) for n in sorted((n for n in names if n.startswith("scale_")), key=lambda s: int(s.split("_")[-1]))], 1),
        _rotation=np.stack([col(n) for n in sorted((n for n in names if n.startswith("rot")), key=lambda s: int(s.split("_")[-1]))], 1),
    )
    if "binding_0" in names:
        out["binding"] = np.asarray(t["binding_0"]).astype(np.int32)
    return out


FLAME_PARAM_KEYS = ("shape", "expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation", "static_offset", "dynamic_offset")


def save_flame_param(path: str, flame_param: Dict[str, np.ndarray]) -> None:
    np.savez(path, **{k: np.asarray(flame_param[k], np.float32) for k in FLAME_PARAM_KEYS if k in flame_param})


def load_flame_param(path: str) -> Dict[str, np.ndarray]:
    """flame_param.npz schema; only float32 arrays are taken, like the reference's motion loader (:239-258)."""
    z = np.load(path)
    return {k: z[k] for k in z.files if z[k].dtype == np.float32}


# -------------------------------------------------------------------------------------------------
# splat ORDER in memory (a loader-side layout choice: the rasterizer's outputs do not depend on it)
# -------------------------------------------------------------------------------------------------
def morton_order(points: np.ndarray, bits: int = 10) -> np.ndarray:
    """Permutation that lists `points` (N,3) along a 3-D Morton (Z-order) curve of their bounding box, `bits` bits per axis.
    Neighbours in memory are then neighbours in space -- and on screen, for every camera: a workgroup that takes a run of consecutive splats
    touches a compact set of tiles, so the binning pass's scattered 8-byte entries of one tile are written close together in time (they
    merge in L2 instead of leaving as one 32-byte sector each) and its LDS histograms see fewer distinct tiles."""
    p = np.asarray(points, np.float64)
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-30) * ((1 << bits) - 1)).astype(np.uint64), 0, (1 << bits) - 1)
    code = np.zeros(len(p), np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return np.argsort(code, kind="stable")


def reorder_splats(arrs: Dict[str, np.ndarray], perm: np.ndarray) -> Dict[str, np.ndarray]:
    """Every per-splat array of a leaf dict (reference names, `binding` included) permuted the same way."""
    n = len(perm)
    return {k: (np.ascontiguousarray(np.asarray(v)[perm]) if v is not None and hasattr(v, "shape") and len(v.shape) and v.shape[0] == n else v)
            for k, v in arrs.items()}


def spatial_sort(arrs: Dict[str, np.ndarray], face_centers: Optional[np.ndarray] = None, return_order: bool = False):
    """The leaf dict with its splats in Morton order of their positions: `_xyz` for an unbound model; for a mesh-bound one (`binding` present)
    the centre of the splat's face on the template mesh (`face_centers` (F,3)), the local offset breaking ties."""
    if arrs.get("binding") is not None and face_centers is not None:
        pos = np.asarray(face_centers, np.float64)[np.asarray(arrs["binding"]).astype(np.int64)] + 1e-3 * np.asarray(arrs["_xyz"], np.float64)
    else:
        pos = np.asarray(arrs["_xyz"], np.float64)
    order = morton_order(pos)
    out = reorder_splats(arrs, order)
    return (out, order) if return_order else out      # order[i] = the input row that became row i
