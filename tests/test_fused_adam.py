"""f-4 (optimizer): FusedAdam against torch.optim.Adam with the reference's settings (scene/gaussian_model.py:163-178:
seven groups with their own learning rates, eps = 1e-15), including the state edits of the reference's densification
(prune / concatenate, scene/gaussian_model.py:300-355) between steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _groups(dev, P=3001, C=5):
    g = torch.Generator().manual_seed(3)
    mk = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g).to(dev))
    specs = [("xyz", (P, 3), 1.6e-4), ("f_dc", (P, 1, 3), 2.5e-3), ("f_rest", (P, 15, 3), 1.25e-4), ("opacity", (P, 1), 0.05),
             ("scaling", (P, 3), 5e-3), ("rotation", (P, 4), 1e-3), ("semantic_feature", (P, 1, C), 1e-3)]
    return [{"params": [mk(*s)], "lr": lr, "name": n} for n, s, lr in specs]


def test_matches_torch_adam_over_steps_and_state_edits():
    from fused_adam import FusedAdam
    dev = "cuda:0"
    ga, gb = _groups(dev), _groups(dev)
    a = torch.optim.Adam(ga, lr=0.0, eps=1e-15)
    b = FusedAdam(gb, lr=0.0, eps=1e-15)
    gen = torch.Generator().manual_seed(11)
    for it in range(1, 8):
        for grp_a, grp_b in zip(a.param_groups, b.param_groups):
            pa, pb = grp_a["params"][0], grp_b["params"][0]
            grad = (torch.randn(pa.shape, generator=gen) * (10.0 ** -(it % 4))).to(dev)
            grad[::7] = 0.0                                # never-visible Gaussians: zero gradient, eps = 1e-15 matters
            pa.grad, pb.grad = grad.clone(), grad.clone()
        if it == 3:
            for opt in (a, b):
                opt.param_groups[0]["lr"] = 8e-5           # update_learning_rate (gaussian_model.py:183-190)
        a.step(); b.step()
        if it == 4:   # prune + append like _prune_optimizer / cat_tensors_to_optimizer: same edit on both
            for opt in (a, b):
                for grp in opt.param_groups:
                    p = grp["params"][0]
                    st = opt.state.pop(p)
                    keep = torch.arange(p.shape[0], device=dev) % 5 != 0
                    extra = torch.zeros((17,) + tuple(p.shape[1:]), device=dev)
                    newp = torch.nn.Parameter(torch.cat([p.detach()[keep], extra + 0.25], 0))
                    st["exp_avg"] = torch.cat([st["exp_avg"][keep], torch.zeros_like(extra)], 0)
                    st["exp_avg_sq"] = torch.cat([st["exp_avg_sq"][keep], torch.zeros_like(extra)], 0)
                    grp["params"][0] = newp
                    opt.state[newp] = st
        for grp_a, grp_b in zip(a.param_groups, b.param_groups):
            pa, pb = grp_a["params"][0].detach(), grp_b["params"][0].detach()
            assert pa.shape == pb.shape
            assert float((pa - pb).abs().max()) <= 2e-6 * float(pa.abs().max()), (it, grp_a["name"])
            sa, sb = a.state[grp_a["params"][0]], b.state[grp_b["params"][0]]
            assert float(sa["step"]) == float(sb["step"])
            for k in ("exp_avg", "exp_avg_sq"):     # fp32 round-off of two equivalent update formulas
                assert float((sa[k] - sb[k]).abs().max()) <= 2e-6 * float(sa[k].abs().max()), (it, grp_a["name"], k)


def test_rejects_cpu_tensors():
    from fused_adam import FusedAdam
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(Exception):
        FusedAdam([p]).step()


def test_visibility_masked_step_touches_only_visible_rows():
    """The opt-in sparse variant (row f-4): visible rows get exactly the dense update, the others keep parameter and
    moments bit for bit."""
    from fused_adam import FusedAdam
    dev = "cuda:0"
    ga, gb = _groups(dev, P=4099), _groups(dev, P=4099)
    a, b = FusedAdam(ga, lr=0.0, eps=1e-15), FusedAdam(gb, lr=0.0, eps=1e-15)
    gen = torch.Generator().manual_seed(5)
    vis = (torch.rand(4099, generator=gen) < 0.4).to(dev)
    for it in range(3):
        for grp_a, grp_b in zip(a.param_groups, b.param_groups):
            grad = torch.randn(grp_a["params"][0].shape, generator=gen).to(dev) * 0.01
            grp_a["params"][0].grad, grp_b["params"][0].grad = grad.clone(), grad.clone()
        before = [(g["params"][0].detach().clone(), {k: v.clone() for k, v in b.state[g["params"][0]].items() if k != "step"} if it else None)
                  for g in b.param_groups]
        a.step()
        b.step(visibility=vis if it == 2 else None)          # two dense steps build state, the third is masked
        if it < 2:
            continue
        for grp_a, grp_b, (p0, st0) in zip(a.param_groups, b.param_groups, before):
            pa, pb = grp_a["params"][0].detach(), grp_b["params"][0].detach()
            assert torch.equal(pa[vis], pb[vis]), grp_a["name"]
            assert torch.equal(pb[~vis], p0[~vis]), grp_a["name"]
            sa, sb = a.state[grp_a["params"][0]], b.state[grp_b["params"][0]]
            for k in ("exp_avg", "exp_avg_sq"):
                assert torch.equal(sa[k][vis], sb[k][vis]) and torch.equal(sb[k][~vis], st0[k][~vis]), (grp_a["name"], k)


@pytest.mark.parametrize("masked", [False, True])
def test_one_launch_and_per_tensor_paths_are_bit_identical(masked):
    """FusedAdam(multi_tensor=True) - ONE launch over the seven tensors (f3dgs_adam_step_multi) - against the per-tensor entry
    point (f3dgs_adam_step[_rows]): same element-wise arithmetic, so every parameter and moment must be equal bit for bit,
    with and without the visibility mask extension."""
    from fused_adam import FusedAdam
    dev = "cuda:0"
    ga, gb = _groups(dev, P=4099, C=32), _groups(dev, P=4099, C=32)
    a, b = FusedAdam(ga, lr=0.0, eps=1e-15, multi_tensor=False), FusedAdam(gb, lr=0.0, eps=1e-15, multi_tensor=True)
    gen = torch.Generator().manual_seed(5)
    vis = (torch.rand(4099, generator=gen) < 0.6).to(dev) if masked else None
    for it in range(4):
        for grp_a, grp_b in zip(a.param_groups, b.param_groups):
            grad = torch.randn(grp_a["params"][0].shape, generator=gen).to(dev) * 1e-2
            grp_a["params"][0].grad, grp_b["params"][0].grad = grad.clone(), grad.clone()
        a.step(visibility=vis); b.step(visibility=vis)
    for grp_a, grp_b in zip(a.param_groups, b.param_groups):
        pa, pb = grp_a["params"][0], grp_b["params"][0]
        assert torch.equal(pa, pb), grp_a["name"]
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(a.state[pa][k], b.state[pb][k]), (grp_a["name"], k)
