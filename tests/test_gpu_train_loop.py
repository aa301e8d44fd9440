"""The reference's training-loop body (train.py:84-149) run twice on the same start, once with the reference's own glue
and once with this repository's modules in every slot they replace:

                          run A (reference glue)                          run B (product modules)
  model                   reference GaussianModel (bytecode)              the same class
  render                  reference render() (bytecode) -> product op     the same
  feature loss            F.interpolate + l1_loss (train.py:100-104)      feature_loss.fused_feature_l1            (f-2)
  step bookkeeping        train.py:132-133 by hand                        train_step.dp_train_step                 (f-1)
  optimizer               torch.optim.Adam (gaussian_model.py:178)        fused_adam.FusedAdam                     (f-4)
  densification           GaussianModel.densify_and_prune                 densify.densify_and_prune                (f-4)

Bit-equality is not expected over iterations (float atomics order in the rasterizer's gradient sums, two Adam
formulations that differ in the last bit, eps = 1e-15); the runs must TRACK each other: same point counts after each
densification to within 1 %, losses within 3 % of each other at every iteration, and the loss must go down.
"""
import importlib.machinery
import importlib.util
import math
import os
import types

import pytest
import torch
import torch.nn.functional as F

import refutil as ru
from test_gpu_dropin import reference_render  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Opt:
    percent_dense = 0.01
    position_lr_init, position_lr_final, position_lr_delay_mult, position_lr_max_steps = 1.6e-4, 1.6e-6, 0.01, 30000
    feature_lr, opacity_lr, scaling_lr, rotation_lr, semantic_feature_lr = 0.0025, 0.05, 0.005, 0.001, 0.001
    lambda_dssim = 0.2
    densify_grad_threshold = 0.00005


def _loss_utils():
    p = os.path.join(ru.REF_DIR, "ref_loss_utils.pyc")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/ref_loss_utils.pyc not built (python oracle/build_ref.py)")
    loader = importlib.machinery.SourcelessFileLoader("ref_loss_utils", p)
    spec = importlib.util.spec_from_loader("ref_loss_utils", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def _start(Model, sc, optimizer=None):
    """The reference's model holding the synthetic scene (raw parameters: log scale, logit opacity, SH split in dc/rest)."""
    m = Model(3)
    par = lambda x: torch.nn.Parameter(x.to(DEV).contiguous().clone().requires_grad_(True))
    m._xyz = par(sc["means3D"])
    m._features_dc, m._features_rest = par(sc["shs"][:, :1]), par(sc["shs"][:, 1:])
    m._scaling, m._rotation = par(torch.log(sc["scales"])), par(sc["rotations"])
    op = sc["opacities"].clamp(1e-4, 1 - 1e-4)
    m._opacity = par(torch.log(op / (1 - op)))
    m._semantic_feature = par(sc["semantic_feature"].reshape(sc["P"], 1, -1))
    m.active_sh_degree = 3
    m.max_radii2D = torch.zeros(sc["P"], device=DEV)
    m.spatial_lr_scale = 1.0
    m.training_setup(_Opt)
    if optimizer is not None:        # the one-line edit of INTEGRATION.md 3.5
        m.optimizer = optimizer(m.optimizer.param_groups, lr=0.0, eps=1e-15)
    return m


def _camera(sc):
    c = types.SimpleNamespace()
    c.FoVx, c.FoVy = 2 * math.atan(sc["tanfovx"]), 2 * math.atan(sc["tanfovy"])
    c.image_height, c.image_width = sc["image_height"], sc["image_width"]
    c.world_view_transform, c.full_proj_transform = sc["viewmatrix"].to(DEV), sc["projmatrix"].to(DEV)
    c.camera_center = sc["campos"].to(DEV)
    return c


def test_training_loop_tracks_the_reference_glue(reference_render):   # noqa: F811
    from synth import make_scene
    import densify
    import train_step
    from feature_loss import fused_feature_l1
    from fused_adam import FusedAdam
    Model = ru.load_reference_gaussian_model()
    lu = _loss_utils()
    sc = make_scene(P=6000, C=16, width=200, height=120, seed=41, yaw_deg=7.0, scale_lo=0.005, scale_hi=0.08)
    cam, bg = _camera(sc), sc["bg"].to(DEV)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    extent = 5.0

    # ground truth: the same scene with perturbed colours / features / positions, rendered once
    g = torch.Generator().manual_seed(7)
    teacher = dict(sc)
    teacher["shs"] = sc["shs"] + 0.15 * torch.randn(sc["shs"].shape, generator=g)
    teacher["semantic_feature"] = sc["semantic_feature"] + 0.3 * torch.randn(sc["semantic_feature"].shape, generator=g)
    teacher["means3D"] = sc["means3D"] + 0.01 * torch.randn(sc["means3D"].shape, generator=g)
    with torch.no_grad():
        t = reference_render(cam, _start(Model, teacher), pipe, bg)
        cam.original_image = t["render"].clone()
        cam.semantic_feature = F.interpolate(t["feature_map"].unsqueeze(0), size=(60, 100), mode="bilinear",
                                             align_corners=True).squeeze(0).clone()

    def loss_reference(pkg, c):                      # train.py:96-106 (no speed-up decoder)
        image, fm = pkg["render"], pkg["feature_map"]
        Ll1 = lu.l1_loss(image, c.original_image)
        fm = F.interpolate(fm.unsqueeze(0), size=c.semantic_feature.shape[1:], mode="bilinear", align_corners=True).squeeze(0)
        return (1.0 - _Opt.lambda_dssim) * Ll1 + _Opt.lambda_dssim * (1.0 - lu.ssim(image, c.original_image)) + lu.l1_loss(fm, c.semantic_feature)

    def loss_product(pkg, c):
        image = pkg["render"]
        Ll1 = lu.l1_loss(image, c.original_image)
        return ((1.0 - _Opt.lambda_dssim) * Ll1 + _Opt.lambda_dssim * (1.0 - lu.ssim(image, c.original_image))
                + fused_feature_l1(pkg["feature_map"], c.semantic_feature, None, None, lowres_grad=True))   # gradient at 60 x 100

    ITER, DENSIFY = 14, (6, 12)
    a, b = _start(Model, sc), _start(Model, sc, FusedAdam)
    la, lb, na, nb = [], [], [], []
    for it in range(1, ITER + 1):
        # ---- run A: the reference's loop body, line by line
        pkg = reference_render(cam, a, pipe, bg)
        loss = loss_reference(pkg, cam)
        loss.backward()
        with torch.no_grad():
            vis, radii = pkg["visibility_filter"], pkg["radii"]
            a.max_radii2D[vis] = torch.max(a.max_radii2D[vis], radii[vis])
            a.add_densification_stats(pkg["viewspace_points"], vis)
            if it in DENSIFY:
                torch.manual_seed(100 + it)
                a.densify_and_prune(_Opt.densify_grad_threshold, 0.005, extent, 20 if it > DENSIFY[0] else None)
            a.optimizer.step()
            a.optimizer.zero_grad(set_to_none=True)
        la.append(float(loss.detach())); na.append(a.get_xyz.shape[0])
        # ---- run B: the product modules
        res = train_step.dp_train_step(reference_render, loss_product, b, [cam], pipe, bg)
        with torch.no_grad():
            if it in DENSIFY:
                torch.manual_seed(100 + it)
                densify.densify_and_prune(b, _Opt.densify_grad_threshold, 0.005, extent, 20 if it > DENSIFY[0] else None)
            b.optimizer.step()
            b.optimizer.zero_grad(set_to_none=True)
        lb.append(float(res.loss)); nb.append(b.get_xyz.shape[0])

    print("loss A", [round(x, 6) for x in la]); print("loss B", [round(x, 6) for x in lb]); print("points", na, nb)
    assert all(math.isfinite(x) for x in la + lb)
    k = DENSIFY[0] - 1                      # before the first densification the loss falls monotonically
    assert la[k - 1] < la[0] and lb[k - 1] < lb[0], (la, lb)
    for x, y in zip(la, lb):
        assert abs(x - y) <= 0.03 * abs(x), (la, lb)
    assert na[-1] != sc["P"], "the densification thresholds of this test no longer select anything"
    for x, y in zip(na, nb):
        assert abs(x - y) <= max(2, 0.01 * x), (na, nb)
    assert isinstance(b.optimizer, FusedAdam) and hasattr(b, "_row_pool")
