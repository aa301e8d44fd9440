"""f-4 (wire format): the point-cloud PLY with semantic_{i} columns, pinned to the reference's own `save_ply`
(tests/golden/reference_ply.npz is the structured array the reference builds, captured by
tests/golden/make_reference_ply_vectors.py) and round-tripped through the reader that mirrors its `load_ply`."""
import os

import numpy as np
import pytest

from util import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "reference_ply.npz")
KEYS = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "semantic_feature")


def test_writer_reproduces_the_reference_records(tmp_path):
    import ply_io
    g = np.load(GOLD)
    args = [g[k] for k in KEYS]
    body = ply_io.pack_vertices(*args)
    assert body.dtype == np.float32 and np.array_equal(body, g["body"])            # bit for bit
    names = ply_io.attribute_names(3, 45, 3, 4, g["semantic_feature"].shape[2])
    assert names == list(g["names"]) == list(g["attribute_list"])
    path = str(tmp_path / "sub" / "point_cloud.ply")
    ply_io.save_ply(path, *args)
    raw = open(path, "rb").read()
    head, payload = raw.split(b"end_header\n", 1)
    lines = head.decode().split("\n")
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0" and lines[2] == f"element vertex {body.shape[0]}"
    assert [l.split()[-1] for l in lines[3:] if l] == names and all(l.startswith("property float ") for l in lines[3:] if l)
    assert payload == g["body"].astype("<f4").tobytes()


def test_reader_round_trip_and_column_order_independence(tmp_path):
    import ply_io
    g = np.load(GOLD)
    path = str(tmp_path / "pc.ply")
    ply_io.save_ply(path, *[g[k] for k in KEYS])
    back = ply_io.load_ply(path)
    for k in KEYS:
        assert back[k].dtype == np.float32 and np.array_equal(back[k], g[k]), k
    # the reference's load_ply addresses columns by NAME (sorted by their numeric suffix): shuffle the columns
    v = ply_io.read_vertices(path)
    order = list(v.dtype.names)
    np.random.default_rng(0).shuffle(order)
    shuffled = str(tmp_path / "shuffled.ply")
    with open(shuffled, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\ncomment shuffled\nelement vertex %d\n" % len(v)).encode())
        f.write("".join(f"property float {n}\n" for n in order).encode() + b"end_header\n")
        f.write(np.stack([v[n] for n in order], axis=1).astype("<f4").tobytes())
    back2 = ply_io.load_ply(shuffled)
    for k in KEYS:
        assert np.array_equal(back2[k], g[k]), k
    with pytest.raises(ValueError):
        ply_io.load_ply(shuffled, max_sh_degree=2)          # wrong number of f_rest columns (gaussian_model.py:250)
