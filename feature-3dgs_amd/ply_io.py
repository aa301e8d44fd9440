"""The Feature-3DGS point-cloud file: a binary little-endian PLY with one float32 property per parameter component,
including `semantic_{i}` columns ("PLY wire format").

Writes byte-for-byte what the reference's `GaussianModel.save_ply` hands to plyfile and reads what its `load_ply`
reads (scene/gaussian_model.py:193-281), without plyfile: property order
    x y z nx ny nz  f_dc_0..2  f_rest_0..44  opacity  scale_0..2  rot_0..3  semantic_0..C-1
with f_dc / f_rest / semantic stored channel-major (`transpose(1, 2).flatten(1)` there): f_rest_{c*15 + k} is
coefficient k+1 of colour channel c, semantic_{i} is feature i.  All tensors are the model's RAW parameters
(pre-activation opacity / scaling), exactly as the reference stores them.
"""
from __future__ import annotations

import os
from typing import Dict

import numpy as np

_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4",
              "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4",
              "float32": "f4", "float64": "f8"}


def attribute_names(n_dc: int, n_rest: int, n_scale: int, n_rot: int, n_semantic: int):
    """Property names in file order (scene/gaussian_model.py:193-209)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)] + ["opacity"]
    names += [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]
    names += [f"semantic_{i}" for i in range(n_semantic)]
    return names


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def pack_vertices(xyz, features_dc, features_rest, opacity, scaling, rotation, semantic_feature) -> np.ndarray:
    """(P, n_properties) float32 matrix in file order from the model's parameter tensors:
    xyz (P,3), features_dc (P,1,3), features_rest (P,K,3), opacity (P,1), scaling (P,3), rotation (P,4),
    semantic_feature (P,1,C)."""
    xyz = _np(xyz).astype(np.float32)
    P = xyz.shape[0]
    chan_major = lambda t: np.ascontiguousarray(np.transpose(_np(t).astype(np.float32), (0, 2, 1))).reshape(P, -1)
    return np.concatenate([xyz, np.zeros_like(xyz), chan_major(features_dc), chan_major(features_rest),
                           _np(opacity).astype(np.float32).reshape(P, -1), _np(scaling).astype(np.float32).reshape(P, -1),
                           _np(rotation).astype(np.float32).reshape(P, -1), chan_major(semantic_feature)], axis=1)


def save_ply(path: str, xyz, features_dc, features_rest, opacity, scaling, rotation, semantic_feature) -> None:
    body = pack_vertices(xyz, features_dc, features_rest, opacity, scaling, rotation, semantic_feature)
    P = body.shape[0]
    names = attribute_names(_np(features_dc).shape[1] * _np(features_dc).shape[2],
                            _np(features_rest).shape[1] * _np(features_rest).shape[2], _np(scaling).shape[1],
                            _np(rotation).shape[1], _np(semantic_feature).shape[1] * _np(semantic_feature).shape[2])
    assert len(names) == body.shape[1]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(body, dtype="<f4").tobytes())


def read_vertices(path: str) -> np.ndarray:
    """The `vertex` element of a binary little-endian (or ascii) PLY as a numpy structured array."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: header has no end_header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif props:
                    break           # elements after `vertex` are not needed
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "binary_little_endian":
            dt = np.dtype([(n, "<" + t) for n, t in props])
            # re-open at the end of the header: the loop above may have stopped early at a later element
            f.seek(0)
            raw = f.read()
            start = raw.index(b"end_header\n") + len(b"end_header\n")
            return np.frombuffer(raw, dtype=dt, count=count, offset=start)
        if fmt == "ascii":
            f.seek(0)
            raw = f.read()
            start = raw.index(b"end_header\n") + len(b"end_header\n")
            flat = np.array(raw[start:].split()[:count * len(props)], dtype=np.float64).reshape(count, len(props))
            out = np.empty(count, dtype=[(n, t) for n, t in props])
            for i, (n, _) in enumerate(props):
                out[n] = flat[:, i]
            return out
        raise ValueError(f"{path}: unsupported PLY format {fmt}")


def load_ply(path: str, max_sh_degree: int = 3) -> Dict[str, np.ndarray]:
    """The model's parameter tensors (float32 numpy, the reference's shapes: scene/gaussian_model.py:233-281)."""
    v = read_vertices(path)
    names = v.dtype.names
    col = lambda n: np.asarray(v[n], np.float32)
    P = v.shape[0]
    by_index = lambda prefix: sorted((n for n in names if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
    xyz = np.stack([col("x"), col("y"), col("z")], axis=1)
    f_dc = np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], axis=1).reshape(P, 3, 1)
    rest = by_index("f_rest_")
    if len(rest) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {len(rest)} f_rest_ columns, expected {3 * (max_sh_degree + 1) ** 2 - 3}")
    f_rest = np.stack([col(n) for n in rest], axis=1).reshape(P, 3, (max_sh_degree + 1) ** 2 - 1) if rest else np.zeros((P, 3, 0), np.float32)
    sem = by_index("semantic_")
    semantic = np.stack([col(n) for n in sem], axis=1).reshape(P, len(sem), 1) if sem else np.zeros((P, 0, 1), np.float32)
    return {
        "xyz": xyz,
        "features_dc": np.ascontiguousarray(np.transpose(f_dc, (0, 2, 1))),            # (P, 1, 3)
        "features_rest": np.ascontiguousarray(np.transpose(f_rest, (0, 2, 1))),        # (P, K, 3)
        "opacity": col("opacity").reshape(P, 1),
        "scaling": np.stack([col(n) for n in by_index("scale_")], axis=1),
        "rotation": np.stack([col(n) for n in by_index("rot")], axis=1),
        "semantic_feature": np.ascontiguousarray(np.transpose(semantic, (0, 2, 1))),   # (P, 1, C)
    }
