"""Generates tests/golden/reference_ply.npz by running the REFERENCE's own `GaussianModel.save_ply`
(scene/gaussian_model.py:193-231) on seeded parameters.  plyfile is not installed here, so a stand-in module captures
the structured array the reference builds (`PlyElement.describe(elements, 'vertex')`) instead of writing it; the
property order, the channel-major flattening and every value are the reference's.

    python tests/golden/make_reference_ply_vectors.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    captured = {}
    ply = types.ModuleType("plyfile")

    class PlyElement:
        @staticmethod
        def describe(arr, name):
            captured["elements"], captured["name"] = arr.copy(), name
            return name

    class PlyData:
        def __init__(self, els):
            pass

        def write(self, path):
            captured["path"] = path
    ply.PlyElement, ply.PlyData = PlyElement, PlyData
    sys.modules["plyfile"] = ply
    sys.path.insert(0, os.path.join(ROOT, "feature-3dgs_amd"))     # simple_knn._C (imported at module level there)
    sys.path.insert(0, REF)
    from scene.gaussian_model import GaussianModel  # noqa: E402  (reference code)

    g = torch.Generator().manual_seed(99)
    P, C = 37, 5
    m = GaussianModel(3)
    m._xyz = torch.randn(P, 3, generator=g)
    m._features_dc = torch.randn(P, 1, 3, generator=g)
    m._features_rest = torch.randn(P, 15, 3, generator=g)
    m._opacity = torch.randn(P, 1, generator=g)
    m._scaling = torch.randn(P, 3, generator=g)
    m._rotation = torch.randn(P, 4, generator=g)
    m._semantic_feature = torch.randn(P, 1, C, generator=g)
    m.save_ply("/tmp/f3dgs_ref_ply/point_cloud.ply")
    el = captured["elements"]
    out = dict(names=np.array(el.dtype.names), body=np.frombuffer(el.tobytes(), dtype="<f4").reshape(P, -1).copy(),
               attribute_list=np.array(m.construct_list_of_attributes()), xyz=m._xyz.numpy(), features_dc=m._features_dc.numpy(),
               features_rest=m._features_rest.numpy(), opacity=m._opacity.numpy(), scaling=m._scaling.numpy(),
               rotation=m._rotation.numpy(), semantic_feature=m._semantic_feature.numpy())
    np.savez_compressed(os.path.join(HERE, "reference_ply.npz"), **out)
    print("wrote reference_ply.npz:", el.dtype.names[:8], "...", len(el.dtype.names), "properties")


if __name__ == "__main__":
    main()
