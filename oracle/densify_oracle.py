"""CPU restatement (numpy) of the reference's densification, step by step - TEST INFRASTRUCTURE ONLY.

Only tests/ may import this file; the product (feature-3dgs_amd/densify.py + csrc/densify.hip) never does.
Parity pin: tests/test_densify.py runs the reference's own `GaussianModel.densify_and_prune` (bytecode compiled from
/root/reference/scene/gaussian_model.py by oracle/build_ref.py) on the GPU box against the same inputs; this
restatement is checked there too and serves as the checker where `oracle/_ref` is absent.

It follows the reference LITERALLY, including the intermediate materialisations the product folds away:

  densify_and_prune      scene/gaussian_model.py:415-431
  densify_and_clone      :398-413      mask -> rows appended               (cat_tensors_to_optimizer :337-357)
  densify_and_split      :378-396      mask over the grown set -> N children appended, the split rows pruned
  densification_postfix  :359-376      statistics reset to zeros
  prune_points           :316-331      mask -> rows removed                (_prune_optimizer :300-314)

State: a dict with the parameters `xyz` (P,3) `f_dc` (P,1,3) `f_rest` (P,K,3) `opacity` (P,1) `scaling` (P,3)
`rotation` (P,4) `semantic_feature` (P,1,C), for each of them optional Adam moments `<name>.exp_avg`,
`<name>.exp_avg_sq`, and the statistics `xyz_gradient_accum` (P,1), `denom` (P,1), `max_radii2D` (P,).
The random draw of `densify_and_split` (:387, torch.normal) is an INPUT here: `normal(std)` must return the samples
for the given (N*Ns, 3) standard deviations.
"""
import numpy as np

PARAMS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "semantic_feature")
f32 = np.float32


def _cat(state, new):            # cat_tensors_to_optimizer (:337-357): rows appended, their moments are zeros
    for n in PARAMS:
        ext = new[n].astype(f32)
        state[n] = np.concatenate((state[n], ext), axis=0)
        for m in (".exp_avg", ".exp_avg_sq"):
            if n + m in state:
                state[n + m] = np.concatenate((state[n + m], np.zeros_like(ext)), axis=0)


def _postfix(state, new):        # densification_postfix (:359-376)
    _cat(state, new)
    P = state["xyz"].shape[0]
    state["xyz_gradient_accum"] = np.zeros((P, 1), f32)
    state["denom"] = np.zeros((P, 1), f32)
    state["max_radii2D"] = np.zeros((P,), f32)


def prune_points(state, mask):   # :316-331 with _prune_optimizer (:300-314)
    keep = ~np.asarray(mask, bool).reshape(-1)
    for n in PARAMS:
        state[n] = state[n][keep]
        for m in (".exp_avg", ".exp_avg_sq"):
            if n + m in state:
                state[n + m] = state[n + m][keep]
    for n in ("xyz_gradient_accum", "denom", "max_radii2D"):
        state[n] = state[n][keep]


def _rotations(r):               # utils/general_utils.py:78-99
    r = r.astype(f32)
    norm = np.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.zeros((q.shape[0], 3, 3), f32)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(f32)))).astype(f32)


def densify_and_clone(state, grads, grad_threshold, scene_extent, percent_dense):   # :398-413
    sel = np.linalg.norm(grads, axis=-1) >= f32(grad_threshold)
    sel &= np.exp(state["scaling"]).max(axis=1) <= f32(percent_dense * scene_extent)
    _postfix(state, {n: state[n][sel] for n in PARAMS})
    return int(sel.sum())


def densify_and_split(state, grads, grad_threshold, scene_extent, percent_dense, normal, N=2):   # :378-396
    P = state["xyz"].shape[0]
    padded = np.zeros((P,), f32)
    padded[:grads.shape[0]] = grads.reshape(-1)
    sel = padded >= f32(grad_threshold)
    scal = np.exp(state["scaling"]).astype(f32)
    sel &= scal.max(axis=1) > f32(percent_dense * scene_extent)
    stds = np.tile(scal[sel], (N, 1))
    samples = np.asarray(normal(stds), f32)
    rots = np.tile(_rotations(state["rotation"][sel]), (N, 1, 1))
    new = {"xyz": np.einsum("nij,nj->ni", rots, samples).astype(f32) + np.tile(state["xyz"][sel], (N, 1)),
           "scaling": np.log(np.tile(scal[sel], (N, 1)) / f32(0.8 * N)).astype(f32),
           "rotation": np.tile(state["rotation"][sel], (N, 1)),
           "f_dc": np.tile(state["f_dc"][sel], (N, 1, 1)),
           "f_rest": np.tile(state["f_rest"][sel], (N, 1, 1)),
           "opacity": np.tile(state["opacity"][sel], (N, 1)),
           "semantic_feature": np.tile(state["semantic_feature"][sel], (N, 1, 1))}
    _postfix(state, new)
    prune_points(state, np.concatenate((sel, np.zeros(N * int(sel.sum()), bool))))
    return int(sel.sum())


def densify_and_prune(state, max_grad, min_opacity, extent, max_screen_size, percent_dense, normal, N=2):   # :415-431
    with np.errstate(divide="ignore", invalid="ignore"):
        grads = (state["xyz_gradient_accum"] / state["denom"]).astype(f32)
    grads[np.isnan(grads)] = 0.0
    cloned = densify_and_clone(state, grads, max_grad, extent, percent_dense)
    split = densify_and_split(state, grads, max_grad, extent, percent_dense, normal, N)
    prune = (_sigmoid(state["opacity"]) < f32(min_opacity)).reshape(-1)
    if max_screen_size:
        big_vs = state["max_radii2D"] > max_screen_size
        big_ws = np.exp(state["scaling"]).max(axis=1) > f32(0.1 * extent)
        prune = prune | big_vs | big_ws
    prune_points(state, prune)
    return {"cloned": cloned, "split": split, "pruned": int(prune.sum()), "points": state["xyz"].shape[0]}
