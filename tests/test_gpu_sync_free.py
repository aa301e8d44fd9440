"""The sync-free forward (option sync_free, include/f3dgs.h) and a step replayed from a HIP graph (graph_step.py).

The reference reads `num_rendered` back in the middle of its forward call (rasterizer_impl.cu:283) and sizes the binning
buffer by it.  With sync_free = 1 this library carves that buffer for a provision, lets the emit kernel and the tile sort read
the count on the device, and reads the count on the host only behind the call's last launch.  What must hold:
  * lists, images and radii BIT-identical to sync_free = 0 - whether the provision was generous, exact or too small (one retry);
  * gradients equal up to the order of the atomic sums;
  * a step captured with torch.cuda.graph replays to the eager step's results, for new contents of its static inputs as well,
    and a replayed frame that does not fit the provision is reported (and repaired by a new capture), never silently wrong.
"""
import numpy as np
import pytest
import torch

from test_gpu_parity import _lib, _raw_forward, _read, _scene
from util import run_hip

pytestmark = pytest.mark.gpu

# (P, W, H, C): the first two stay inside the one-launch LDS-resident sorts (<= 16384 pairs), the third takes the
# three-kernel passes for both sorts
SCENES = {"small": dict(P=3000, width=96, height=64, C=0, seed=21, scale_lo=0.005, scale_hi=0.05),
          "small-features": dict(P=2500, width=112, height=80, C=16, seed=22, scale_lo=0.005, scale_hi=0.05),
          "mid": dict(P=30000, width=320, height=200, C=32, seed=23, scale_lo=0.005, scale_hi=0.06)}


def _lists(scene):
    lib = _lib()
    res = _raw_forward(scene)
    from diff_gaussian_rasterization import _C
    counts = _C.forward_counts()
    n_own = int(_read(lib, "counters", scene, res, np.uint32, 16)[0])
    tiles = ((scene["image_width"] + 15) // 16) * ((scene["image_height"] + 15) // 16)
    return dict(n=res[0], n_own=n_own, counts=counts,
                point_list=_read(lib, "point_list", scene, res, np.uint32, n_own),
                tile_sorted=_read(lib, "tile_sorted", scene, res, np.uint32, n_own),
                ranges=_read(lib, "ranges", scene, res, np.uint32, 2 * tiles),
                n_contrib=_read(lib, "n_contrib", scene, res, np.uint32, scene["image_width"] * scene["image_height"]),
                color=res[1].cpu().numpy(), feat=res[2].cpu().numpy(), depth=res[3].cpu().numpy(), radii=res[4].cpu().numpy())


def _same_lists(a, b):
    assert a["n"] == b["n"] and a["n_own"] == b["n_own"]
    for k in ("point_list", "tile_sorted", "ranges", "n_contrib", "color", "feat", "depth", "radii"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("name", sorted(SCENES))
@pytest.mark.parametrize("provision", ["auto", "generous", "exact", "one-short", "tiny"])
def test_sync_free_forward_is_bit_identical(name, provision, option):
    """Whatever the provision: the same lists, ranges, images and radii as the blocking forward; the host's words say what the
    call provided for (a provision that was too small is replaced by the exact length in the second round of the call)."""
    scene = _scene(**SCENES[name])
    option("sync_free", 0)
    want = _lists(scene)
    n = want["n_own"]
    assert n > 0 and want["counts"][3] == n
    option("sync_free", 1)
    cap = {"auto": 0, "generous": 4 * n + 1000, "exact": n, "one-short": n - 1, "tiny": 64}[provision]
    option("instance_capacity", cap)
    got = _lists(scene)
    _same_lists(got, want)
    c = got["counts"]
    assert c[0] == n and c[1] == want["n"] and c[4] == 0
    if provision == "auto":
        assert c[3] == n + n // 4 + 4096          # (1.25 x the count this thread read last - the blocking call above - + 4096)
    elif provision in ("generous", "exact"):
        assert c[3] == cap
    else:
        assert c[3] == n                           # the retry carved for the exact length
    # and once more: the buffers of the previous frame are gone, the provision comes from the count just read
    got2 = _lists(scene)
    _same_lists(got2, want)


@pytest.mark.parametrize("name", ["small-features", "mid"])
def test_sync_free_step_has_the_blocking_steps_gradients(name, option):
    scene = _scene(**SCENES[name])
    option("sync_free", 0)
    o0, g0 = run_hip(scene)
    _o, g0b = run_hip(scene)
    option("sync_free", 1)
    option("instance_capacity", 17)        # every call goes through the retry
    o1, g1 = run_hip(scene)
    option("instance_capacity", 0)
    o2, g2 = run_hip(scene)
    for o in (o1, o2):
        for k in ("color", "feature_map", "depth", "radii"):
            assert np.array_equal(o[k], o0[k]), k
    for g in (g1, g2):
        for k, v in g0.items():
            if v is None:
                continue
            scale = np.abs(v).max() + 1e-30
            noise = np.abs(g0b[k] - v).max() / scale
            assert np.abs(g[k] - v).max() / scale <= 4 * noise + 1e-5, k


def _static_step(scene, dev="cuda:0"):
    """A forward + backward of the op on STATIC tensors (what a graph needs): returns (fn, inputs, outputs, grads)."""
    import diff_gaussian_rasterization as dgr
    t = lambda x: x.to(dev)
    P = scene["means3D"].shape[0]
    settings = dgr.GaussianRasterizationSettings(
        image_height=scene["image_height"], image_width=scene["image_width"], tanfovx=scene["tanfovx"],
        tanfovy=scene["tanfovy"], bg=t(scene["bg"]), scale_modifier=scene["scale_modifier"],
        viewmatrix=t(scene["viewmatrix"]), projmatrix=t(scene["projmatrix"]), sh_degree=scene["sh_degree"],
        campos=t(scene["campos"]), prefiltered=False, debug=False)
    names = ("means3D", "opacities", "semantic_feature", "shs", "scales", "rotations")
    L = {k: t(scene[k]).clone().requires_grad_(True) for k in names}
    L["means2D"] = torch.zeros(P, 3, device=dev, requires_grad=True)
    up = {k: t(scene[k]) for k in ("dL_dcolor", "dL_ddepth", "dL_dfeature")}
    grads = {k: torch.zeros_like(v) for k, v in L.items()}
    outs = {}
    rast = dgr.GaussianRasterizer(settings)

    def fn():
        color, feat, radii, depth = rast(**L)
        loss = (color * up["dL_dcolor"]).sum() + (depth * up["dL_ddepth"]).sum()
        if scene["C"]:
            loss = loss + (feat * up["dL_dfeature"]).sum()
        gs = torch.autograd.grad(loss, [L[k] for k in grads], allow_unused=True)
        for k, g in zip(grads, gs):
            if g is not None and g.numel():
                grads[k].copy_(g)
        outs["color"], outs["feat"], outs["depth"], outs["radii"] = color.detach(), feat.detach(), depth.detach(), radii
        return loss.detach()
    return fn, L, outs, grads


def _snapshot(outs, grads):
    torch.cuda.synchronize()
    return ({k: v.cpu().numpy().copy() for k, v in outs.items()}, {k: v.cpu().numpy().copy() for k, v in grads.items()})


def _close(got, want, again, what):
    for k, v in want.items():
        if v.size == 0:
            continue
        scale = np.abs(v).max() + 1e-30
        noise = np.abs(again[k] - v).max() / scale
        assert np.abs(got[k] - v).max() / scale <= 4 * noise + 1e-5, (what, k)      # (1e-5 max|g|: the absolute part of the gradient bound)


@pytest.mark.parametrize("name", ["small", "small-features", "mid"])
def test_captured_step_replays_the_eager_step(name, option):
    from graph_step import CapturedStep
    from diff_gaussian_rasterization import _C
    option("bwd_bf16", 1)        # one contraction on both sides (what the default takes inside a capture: the test below)
    scene = _scene(**SCENES[name])
    fn, L, outs, grads = _static_step(scene)
    fn()
    o_want, g_want = _snapshot(outs, grads)
    fn()
    _o, g_again = _snapshot(outs, grads)
    # list lengths of the three contents this test replays (scales x 1, x 1.08, x 2.16): the provision sits between the last two
    n_entries = []
    base = L["scales"].detach().clone()
    for f in (1.0, 1.08, 2.16, 1.0):
        with torch.no_grad():
            L["scales"].copy_(base * f)
        fn()
        torch.cuda.synchronize()
        n_entries.append(_C.forward_counts()[0])
    assert n_entries.pop() == n_entries[0]
    assert n_entries[0] < n_entries[1] < n_entries[2] - 1, n_entries
    step = CapturedStep(fn, capacity=(n_entries[1] + n_entries[2]) // 2).capture()
    assert _C.get_option("sync_free") == -1 and _C.get_option("instance_capacity") == 0, "the capture leaves the caller's options as they were"
    graph_outs = dict(outs)          # the tensors the graph writes (an eager call of fn puts new ones into `outs`)
    for v in grads.values():
        v.zero_()
    for _ in range(3):
        step.replay()
    assert step.check() and step.captures == 1
    o_got, g_got = _snapshot(graph_outs, grads)
    for k in o_want:
        assert np.array_equal(o_got[k], o_want[k]), k
    _close(g_got, g_want, g_again, "replay")
    c = step.counts()
    assert c[0] == n_entries[0] and c[3] == (n_entries[1] + n_entries[2]) // 2 and c[4] == 0

    # new contents of the static inputs, same graph: every Gaussian eight percent larger - more list entries, inside the provision
    with torch.no_grad():
        L["scales"].mul_(1.08)
    fn()
    o_want2, g_want2 = _snapshot(outs, grads)
    fn()
    _o, g_again2 = _snapshot(outs, grads)
    step.replay()
    assert step.check() and step.captures == 1
    o_got2, g_got2 = _snapshot(graph_outs, grads)
    assert step.counts()[0] == n_entries[1]
    for k in o_want2:
        assert np.array_equal(o_got2[k], o_want2[k]), k
    _close(g_got2, g_want2, g_again2, "replay of new contents")

    # ... and twice as large: the lists outgrow the provision; the replayed frame is void and SAYS so, check() captures again
    with torch.no_grad():
        L["scales"].mul_(2.0)
    fn()
    o_want3, g_want3 = _snapshot(outs, grads)
    fn()
    _o, g_again3 = _snapshot(outs, grads)
    step.replay()
    assert step.counts()[4] == 1 and step.counts()[0] == n_entries[2] > step.counts()[3]
    assert step.check() is False and step.captures == 2
    o_got3, g_got3 = _snapshot(outs, grads)         # (the new capture ran fn: `outs` holds the new graph's tensors)
    for k in o_want3:
        assert np.array_equal(o_got3[k], o_want3[k]), k
    _close(g_got3, g_want3, g_again3, "replay after the re-capture")
    assert step.check() and step.captures == 2


@pytest.mark.parametrize("kind", ["benign", "needles"])
def test_a_captured_backward_picks_its_contraction_on_the_device(kind):
    """bwd_bf16 = -1 (default) chooses the contraction of the blend backward by a word the HOST reads from the frame; a captured
    frame is never read.  Its backward launches BOTH shapes of the first window - bf16 two-term and hybrid (moment block in exact
    fp32) - and the frame's long-axis word lets exactly one of them run: the replayed gradients are the eager default's (which
    took the bf16 shape on the benign scene, the hybrid on the needles) up to the order of the sums."""
    from graph_step import CapturedStep
    from diff_gaussian_rasterization import _C
    from test_gpu_parity import _needle_scene
    assert _C.get_option("bwd_bf16") == -1
    scene = _scene(**SCENES["small-features"]) if kind == "benign" else _needle_scene(64, "needle", P=4000, C=16)
    fn, _L, _outs, grads = _static_step(scene)
    fn()
    torch.cuda.synchronize()
    assert _C.last_backward_contraction() == (1 if kind == "benign" else 2)
    _o, g_want = _snapshot({}, grads)
    fn()
    _o, g_again = _snapshot({}, grads)
    step = CapturedStep(fn).capture()
    assert _C.last_backward_contraction() == 3
    for _ in range(2):
        step.replay()
    assert step.check() and step.counts()[2] == (0 if kind == "benign" else 1)
    _o, g_got = _snapshot({}, grads)
    for k, v in g_want.items():
        if v.size == 0:
            continue
        bound = 1e-3 * np.abs(v) + 1e-5 * np.abs(v).max()
        noise = float((np.abs(g_again[k] - v) / bound).max())
        worst = float((np.abs(g_got[k] - v) / bound).max())
        # the same kernel on the same sums: apart as far as two eager runs are (on needles the covariance chain spreads those)
        slack = 1.0 if (kind == "needles" and k in ("scales", "rotations")) else 0.25
        assert worst <= slack + 4.0 * noise, (k, worst, noise)


def test_the_default_is_the_blocking_read_outside_a_capture():
    """sync_free = -1 (default): an eager call carves the binning buffer for exactly the count it read (measured: eager, the
    sync-free path buys nothing - tools/sync_free_ab.py); a capture takes the sync-free path (test_a_captured_backward... run under it)."""
    from diff_gaussian_rasterization import _C
    assert _C.get_option("sync_free") == -1
    scene = _scene(**SCENES["mid"])
    for _ in range(2):
        a = _lists(scene)
    assert a["counts"][3] == a["n_own"]


def test_a_capture_without_the_option_is_refused_loudly(option):
    """sync_free = 0: the forward call must read the count on the host - inside a capture that is an error with a message, not a
    hang or a broken graph."""
    option("sync_free", 0)
    scene = _scene(**SCENES["small"])
    fn, _L, _outs, _grads = _static_step(scene)
    fn()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    caught = None
    with torch.cuda.stream(s):
        g.capture_begin()
        try:
            fn()
        except Exception as e:      # noqa: BLE001 (the binding raises RuntimeError; anything else fails the match below)
            caught = e
        finally:
            try:
                g.capture_end()
            except Exception:       # noqa: BLE001 (an empty or abandoned capture may not instantiate; not what is under test)
                pass
    assert caught is not None and "sync_free" in str(caught), caught
    del g
    torch.cuda.synchronize()
    fn()        # the library is still usable
    torch.cuda.synchronize()
