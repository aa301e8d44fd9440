/*
 * f3dgs.h — C ABI of the MI355X-native differentiable Gaussian rasterizer
 * (libf3dgs_hip.so).
 *
 * This is the drop-in boundary for the one hot path of Feature-3DGS: the
 * tile-based forward/backward rasterizer.  Every entry point below replaces
 * one static method of the reference's C++ class `CudaRasterizer::Rasterizer`
 *   (reference: submodules/diff-gaussian-rasterization-feature/
 *               cuda_rasterizer/rasterizer.h:20-94)
 * which is what the reference's torch binding (rasterize_points.cu:35-236)
 * calls.  Signatures use plain pointers and sizes only: no torch types, no
 * C++ types.  All pointers are DEVICE pointers unless stated otherwise; all
 * float tensors are contiguous fp32; matrices are the 16-float
 * "transposed" matrices the reference passes (scene/cameras.py:55-58), i.e.
 * element (row r, col c) of the mathematical matrix is m[4*c + r].
 *
 * Differences to the reference interface (all deliberate):
 *   - `C` (number of semantic-feature channels) is a RUN-TIME argument.  The
 *     reference bakes it in as the macro NUM_SEMANTIC_CHANNELS
 *     (cuda_rasterizer/config.h:16) and must be recompiled per feature dim.
 *     C == 0 is valid.
 *   - The three `std::function<char*(size_t)>` resize hooks
 *     (rasterizer.h:36-38) become (function pointer, context) pairs.
 *   - Every call takes the HIP stream to enqueue on (the reference uses the
 *     legacy default stream) and returns an int status instead of throwing.
 *   - Optional inputs follow the reference's convention: NULL means "absent"
 *     (forward.cu:204,240; backward.cu:398,402).
 *   - backward() zero-initialises / fully overwrites every output itself; the
 *     caller may pass uninitialised memory (the reference requires zeroed
 *     buffers, rasterize_points.cu:163-173).
 *
 * Alignment: the per-Gaussian kernels read rotations, SH rows and feature
 * rows, and write their gradients, with 16-byte accesses.  `shs`,
 * `rotations`, `semantic_feature`, `dL_dsh`, `dL_drot`, `dL_dconic`,
 * `dL_dsemantic_feature` and `scratch` must therefore be 16-byte aligned
 * (hipMalloc and torch allocations are; a view at an odd storage offset is
 * not).  f3dgs_forward / f3dgs_backward return F3DGS_ERR_INVALID_ARGUMENT for
 * a misaligned pointer; the torch binding copies such a view first.  Image
 * planes may have any alignment (aligned ones take the vector path).
 *
 * Status codes: 0 = ok, negative = error; f3dgs_last_error() returns a
 * thread-local human-readable message for the last failing call.
 */
#ifndef F3DGS_H_INCLUDED
#define F3DGS_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F3DGS_OK 0
#define F3DGS_ERR_INVALID_ARGUMENT (-1)
#define F3DGS_ERR_HIP (-2)
#define F3DGS_ERR_ALLOC (-3)
#define F3DGS_ERR_UNSUPPORTED (-4)

/* Resize hook: must return a device pointer to at least `nbytes` bytes that
 * stays valid until the matching backward call has completed.  Mirrors
 * `std::function<char*(size_t)>` of rasterizer.h:36-38 /
 * resizeFunctional() of rasterize_points.cu:27-33. */
typedef char* (*f3dgs_resize_fn)(void* ctx, size_t nbytes);

/* Library / ABI version: major*10000 + minor*100 + patch (3.5.0 -> 30500). */
int f3dgs_version(void);

/* Thread-local message of the last error raised on this host thread. */
const char* f3dgs_last_error(void);

/*
 * Process-wide options (no counterpart in the reference).  Each option selects between complete code paths that
 * are exercised by the test suite (tests/test_abi_and_surface.py fails on an option without a test); none removes
 * work.  Defaults are seeded ONCE, at first use of the library, from the environment variable
 * F3DGS_<NAME IN CAPITALS>; afterwards only these calls change them (no launch path reads the environment).
 *   "tile_cull"      1 (default): instances whose 1/255-alpha ellipse misses a tile are never emitted (they
 *                    could not blend at any of its pixels, so outputs are unchanged); 0: the reference's
 *                    bounding-rectangle lists, bit-identical private state (used by the parity tests)
 *   "feature_mfma"   1 (default): feature contraction of the blend kernels on the matrix pipe (exact fp32);
 *                    0: vector pipe only
 *   "profile"        1: per-stage HIP events, see f3dgs_profile_read; 2: only around the two blend kernels
 *   "sort_onesweep"  0 (default): three-kernel radix passes; 1: single-pass radix scatter with decoupled look-back
 *                    (measured slower on MI355X, kept as a tested alternative)
 *   "bwd_pl"         blend backward formulation: 1 pixel-lane kernel (all sums on the matrix pipe), 0 instance-lane
 *                    kernel, -1 (default) by channel count (pixel-lane from the first feature channel on; from five with
 *                    bwd_bf16 = 0)
 *   "bwd_half"       instance-lane blend backward: 1 (default) chunks of 32 instances against two pixel halves,
 *                    0 chunks of 64
 *   "bwd_order"      blend backward: 1 (default) workgroups take the tiles longest walk first
 *   "bwd_bf16"       pixel-lane blend backward: 1 every per-Gaussian sum contracts on bf16 matrix instructions
 *                    (v_mfma_f32_16x16x32_bf16, fp32 accumulation) with each fp32 operand split into two bf16 terms -
 *                    a relative error of a few 1e-6 per product against the 1e-3 tolerance of the gradients; 0: exact-fp32
 *                    matrix instructions (16x slower per multiply-add, and they block the vector pipe of their SIMD);
 *                    -1 (default): by the frame - bf16 while no visible Gaussian is longer than "bwd_bf16_max_ratio" times
 *                    its width (3D scales and screen-space footprint; f3dgs_forward notes per geometry buffer whether
 *                    the frame holds such a Gaussian - against the ratio in force at that call -, f3dgs_backward looks it
 *                    up), otherwise - and when the note is gone - the HYBRID first window: its moment block (the six
 *                    geometric sums) on exact-fp32 matrix instructions, feature and colour blocks on bf16.  The covariance
 *                    chain behind the blend amplifies an error of the MOMENT sums by the square of that ratio
 *                    (measured: the bf16 shape within a third of the gradient bound up to 16, outside it from 32 on);
 *                    a frame replayed from a graph launches both shapes and the frame's word lets one run on the device;
 *                    the later channel windows of wide features (C > 32: feature sums only, nothing amplifies them) stay on
 *                    the bf16 contraction under -1 whatever the first window took
 *   "bwd_bf16_max_ratio"  (default 16) the axis ratio up to which bwd_bf16 = -1 takes the bf16 contraction
 *   "bwd_wide8"      bf16 shape of the pixel-lane blend backward, feature widths above 96: 1 (default) later channel windows of
 *                    up to 128 channels on eight waves per tile (four of them evaluate the blend weights, all eight contract)
 *                    where more than 64 channels remain - half as many re-evaluations of the lists; 0: 64 channels on four
 *   "bwd_m44"        fp32 shape of the pixel-lane blend backward (bwd_bf16 = 0): 1 (default) the colour / depth sums contract on 4 x 4 matrix blocks
 *                    (v_mfma_f32_4x4x1_16B_f32) instead of a 16-column block of which four are used
 *   "bwd_split16"    fp32 shape of the pixel-lane blend backward (bwd_bf16 = 0) with up to 16 feature channels, and its later channel windows of up to 32: 1 (default)
 *                    the column blocks are split over the four waves by quadrants as well (no wave without matrix work, partial
 *                    sums added in the flush), 0 by columns only
 *   "fwd_wide"       blend forward: 1 (default) 128-channel windows where more than 64 channels remain
 *   "fwd_solo"       blend forward: 1 (default) one 64-thread workgroup per quadrant wave
 *   "sync_free"      0: f3dgs_forward waits for the instance count where the reference does
 *                    (rasterizer_impl.cu:283; here behind the enqueue of the depth sort) and carves the binning buffer for
 *                    exactly that length.  -1 (default): as 1 inside a graph capture, as 0 otherwise (eager, the wait is already
 *                    hidden behind the depth sort: measured, the option changes nothing there).  1: SYNC-FREE FORWARD - the binning buffer is carved for a provision
 *                    ("instance_capacity"), the emit kernel and the tile sort read the count on the device, and the host reads
 *                    it only behind the last launch of the call (it has long been final by then); a frame that found no room
 *                    runs its binning and blend once more with the exact length before the call returns (binning_resize is
 *                    then called a SECOND time in the same forward call, while the first round's kernels may still be in
 *                    flight on the stream: a hook that frees the old buffer must do so in stream order).  Results are
 *                    bit-identical to sync_free = 0.  With sync_free = 1 or -1 the call may also run on a stream that is being
 *                    CAPTURED into a HIP graph (hipStreamBeginCapture / torch.cuda.graph): then nothing is read on the host,
 *                    *num_rendered receives the last count this thread read on the device (at least 1), the blend backward of
 *                    bwd_bf16 = -1 launches both shapes of its first window and the frame's long-axis word lets one of them
 *                    run on the device, and a replayed frame that finds no room is
 *                    VOID: it raises word [4] of f3dgs_forward_counts(), which the owner of the graph checks after a replay
 *                    (and captures again with more room).  Capturing needs one eager forward call on the same host thread and
 *                    device beforehand (pinned count words, a provision); "debug", "profile" and "sort_onesweep" are not
 *                    available inside a capture.
 *   "instance_capacity"  sync_free = 1: entries of the instance lists to provide for; 0 (default): 1.25 x the last count this
 *                    thread read on the device + 4096
 * Unknown names return F3DGS_ERR_INVALID_ARGUMENT.
 */
int f3dgs_set_option(const char* name, int value);
int f3dgs_get_option(const char* name, int* value /* host pointer, out */);

/* Which contraction the last f3dgs_backward of this PROCESS (any thread: PyTorch runs the backward pass on an autograd thread)
 * ran its blend stage with: 1 the pixel-lane kernel in its two-term bf16 shape, 2 its hybrid shape (first window: feature and
 * colour blocks on bf16, the moment block on exact-fp32 matrix instructions), 3 one of those two chosen ON THE DEVICE by the
 * frame's long-axis word (a frame captured into a graph), 0 an exact-fp32 shape (pixel-lane fp32 or the instance-lane kernel),
 * -1 no backward call yet.  A diagnostic for tests and benchmarks. */
int f3dgs_last_backward_contraction(void);
/* The pinned host words the most recent f3dgs_forward of this THREAD reports its frame in (kernel-written unless said otherwise;
 * NULL before the first call): [0] entries of the instance lists, [1] the reference's num_rendered, [2] != 0: a visible Gaussian
 * has a long axis (option bwd_bf16_max_ratio), [3] entries the binning buffer was carved for (written by the host at enqueue),
 * [4] sticky - an emit wave of a CAPTURED frame found no room (the caller clears it).  They are final once the frame's work has
 * completed (synchronise first); a replayed graph writes the words of the call it was captured from. */
const uint32_t* f3dgs_forward_counts(void);
/* ONE view split over several GPUs by tile rows (no counterpart in the reference; SURVEY.md 8(e), "alternative for single huge
 * views"): from this call on the f3dgs_forward calls of this THREAD list and blend only the tiles of rows [tile_row_begin,
 * tile_row_end) of the 16 x 16 tile grid (clipped to the grid; (0, 0) restores the whole view; any other pair with begin >= end
 * is an EMPTY band - the share of a rank beyond the last tile row).  Inside the band the images,
 * final T and n_contrib are bit-identical to the whole view's; outside it the outputs hold the background / zeros.  A Gaussian
 * whose rectangle misses the band is invisible to the call (radius 0, no gradient); *num_rendered counts the band's tiles: over
 * a partition of the rows the counts add up to the whole view's, the element-wise maximum of the radii is the whole view's,
 * and the SUM of the band calls' gradients (each given the upstream gradient of its own rows) is the whole view's gradient -
 * the same all-reduce a view-sharded step ends with (dp.py: band_rows, gather_bands).  The matching f3dgs_backward needs
 * nothing: it reads the band's lists from the forward call's buffers. */
void f3dgs_set_tile_band(int tile_row_begin, int tile_row_end);
/* Host-only diagnostic (no GPU needed): the order in which the blend kernels' workgroups take the tiles of a gx x gy grid when
 * rows [begin, end) are listed - tiles_out[v] = tile of virtual id v (gx * gy entries).  A band is a contiguous run of tile ids,
 * i.e. one XCD's share under the whole-view mapping; with a band the ids are relabelled so that every XCD's run of virtual ids
 * starts with an eighth of the band's tiles.  Returns 1 if the relabelling is on, 0 for the identity (whole view, a band of
 * more than about half the grid, a tiny grid), < 0 on error. */
int f3dgs_debug_band_order(int gx, int gy, int tile_row_begin, int tile_row_end, uint32_t* tiles_out /* host, gx * gy */);
/* Enumeration: the name of option `index` (0, 1, ...), NULL past the end. */
const char* f3dgs_option_name(int index);

/*
 * Replaces CudaRasterizer::Rasterizer::markVisible (rasterizer.h:24-29,
 * rasterizer_impl.cu:141-153): present[i] = (view * p_i).z > 0.2.
 * `present` is a device array of P bytes (0/1).
 */
int f3dgs_mark_visible(
    int P,
    const float* means3D,
    const float* viewmatrix,
    const float* projmatrix,
    uint8_t* present,
    void* stream /* hipStream_t */);

/*
 * Replaces CudaRasterizer::Rasterizer::forward (rasterizer.h:31-62,
 * rasterizer_impl.cu:198-342).
 *
 *   P  number of Gaussians, D active SH degree (0..3), M SH coefficients per
 *      colour channel held by `shs` (0 if absent), C semantic channels.
 *   shs (P,M,3) | colors_precomp (P,3): exactly one non-NULL.
 *   scales (P,3) + rotations (P,4) | cov3D_precomp (P,6): exactly one set.
 *   semantic_feature (P,C) (may be NULL iff C == 0), opacities (P).
 *   out_color (3,H,W), out_feature_map (C,H,W), out_depth (1,H,W), radii (P)
 *   are fully written (radii may be NULL).
 *
 * The three opaque state buffers are sized through the resize hooks and must
 * be handed back unchanged to f3dgs_backward.  Their layout is private to
 * this library.  *num_rendered receives the reference's return value: the
 * number of (tile, Gaussian) instances of the 3-sigma bounding rectangles
 * (rasterizer_impl.cu:279-283).  The private instance lists may be shorter:
 * unless option "tile_cull" is 0, instances whose 1/255-alpha ellipse misses
 * the tile are never emitted (they could not blend at any pixel, so outputs
 * are unchanged).  Reading the counts costs one
 * 8-byte device-to-host copy + stream sync, like rasterizer_impl.cu:283.
 */
int f3dgs_forward(
    f3dgs_resize_fn geometry_resize, void* geometry_ctx,
    f3dgs_resize_fn binning_resize, void* binning_ctx,
    f3dgs_resize_fn image_resize, void* image_ctx,
    int P, int D, int M, int C,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* semantic_feature,
    const float* opacities,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* cam_pos,
    float tan_fovx, float tan_fovy,
    int prefiltered,
    float* out_color,
    float* out_feature_map,
    float* out_depth,
    int* radii,
    int debug,
    void* stream /* hipStream_t */,
    int* num_rendered /* host pointer, out */);

/* Bytes of device scratch f3dgs_backward needs for P Gaussians (the
 * reference cudaMalloc's its scratch inside the call,
 * rasterizer_impl.cu:402-430; here the caller owns it). */
size_t f3dgs_backward_scratch_bytes(int P, int C);

/*
 * Replaces CudaRasterizer::Rasterizer::backward (rasterizer.h:64-94,
 * rasterizer_impl.cu:347-461).
 *
 *   R = num_rendered returned by the matching forward call; geom/binning/
 *   image buffers are the ones that call filled.
 *   dL_dpix (3,H,W), dL_dfeaturepix (C,H,W), dL_depths (1,H,W): upstream
 *   gradients.
 *   Outputs (all fully overwritten): dL_dmean2D (P,3) [x,y in NDC units, z=0],
 *   dL_dopacity (P), dL_dcolor (P,3), dL_dsemantic_feature (P,C),
 *   dL_dmean3D (P,3), dL_dcov3D (P,6), dL_dsh (P,M,3), dL_dscale (P,3),
 *   dL_drot (P,4).  dL_dconic (P,4) and dL_dz (P) are OPTIONAL diagnostics
 *   (NULL = not wanted); the reference allocates but never returns them.
 *   dL_dsh may be NULL iff M == 0; dL_dscale/dL_drot may be NULL iff scales
 *   is NULL.  `scratch` points to f3dgs_backward_scratch_bytes(P, C) bytes.
 */
int f3dgs_backward(
    int P, int D, int M, int C, int R,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* semantic_feature,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* campos,
    float tan_fovx, float tan_fovy,
    const int* radii,
    const char* geom_buffer,
    const char* binning_buffer,
    const char* image_buffer,
    const float* dL_dpix,
    const float* dL_dfeaturepix,
    const float* dL_depths,
    float* dL_dmean2D,
    float* dL_dconic,
    float* dL_dopacity,
    float* dL_dcolor,
    float* dL_dsemantic_feature,
    float* dL_dmean3D,
    float* dL_dcov3D,
    float* dL_dsh,
    float* dL_dscale,
    float* dL_drot,
    float* dL_dz,
    void* scratch,
    int debug,
    void* stream /* hipStream_t */);

/*
 * Fused feature-map loss around the rasterizer (no single counterpart in the reference: it replaces the three
 * PyTorch calls of train.py:99-105 - F.interpolate(bilinear, align_corners=True) to the ground truth's size, the
 * optional 1x1-conv decoder `CNN_decoder` (models/networks.py:107-119) and l1_loss (utils/loss_utils.py:17-18) - and
 * their autograd backward).  feature_map (C,H,W); gt (Cout,Hg,Wg); weight (Cout,C) + bias (Cout) or both NULL for
 * "no decoder" (then Cout must equal C).  Outputs: *loss (device scalar) = mean |decode(resize(feature_map)) - gt|,
 * d_feature_map (C,H,W), d_weight (Cout,C), d_bias (Cout): the gradients of that loss (upstream gradient 1; they
 * scale linearly).  With a decoder C must be 32, 64 or 128 (the contraction runs on the fp32 matrix pipe in
 * 32-channel blocks); other shapes return F3DGS_ERR_UNSUPPORTED.  `scratch`: f3dgs_feature_l1_scratch_bytes(...) bytes.
 *
 * d_feature_map may be NULL: the dense (C,H,W) gradient - zeros at every pixel the resize does not sample, 8 of 9 when the
 * ground truth is a third of the image - is then not written.  The gradient at the LOSS's resolution stays in `scratch`
 * ((Hg*Wg, C) floats, pixel-major, f3dgs_feature_l1_lowres_grad) for as long as the caller keeps `scratch`, and goes to the
 * blend backward through f3dgs_set_feature_grad_lowres, which applies the transposed resize tile by tile.
 */
size_t f3dgs_feature_l1_scratch_bytes(int C, int Cout, int Hg, int Wg, int has_decoder);
int f3dgs_feature_l1(int C, int H, int W, int Cout, int Hg, int Wg, const float* feature_map, const float* weight,
                     const float* bias, const float* gt, float* loss, float* d_feature_map, float* d_weight,
                     float* d_bias, void* scratch, void* stream /* hipStream_t */);

const float* f3dgs_feature_l1_lowres_grad(int C, int Cout, int Hg, int Wg, int has_decoder, void* scratch);

/*
 * The next f3dgs_backward call of THIS thread takes its feature-map gradient (the argument dL_dfeaturepix of
 * rasterize_points.cu:121 `RasterizeGaussiansBackwardCUDA`) at the resolution of the loss: gx = (Hg*Wg, C) floats,
 * pixel-major, dL/d(resized feature map) as f3dgs_feature_l1 leaves it; `scale`: device scalar multiplied in (the upstream
 * gradient of the loss) or NULL.  The blend backward computes, per tile, what F.interpolate's backward would have written
 * there (same products, same order: bit-identical to the dense path) - it reads Hg*Wg*C floats instead of H*W*C.
 * dL_dfeaturepix of that call may be NULL; if it is not, the two are added.  Needs Hg <= H and Wg <= W (shrinking: at
 * most two output samples per source row / column) and option feature_mfma = 1, else the call returns
 * F3DGS_ERR_UNSUPPORTED.  The setting is consumed by that call whatever it returns - including its early returns (P == 0,
 * an argument error) -; gx == NULL clears it.  The kernel reads exactly the Hg*Wg*C floats of `gx`, nothing beyond them.
 */
int f3dgs_set_feature_grad_lowres(const float* gx, int Hg, int Wg, const float* scale);

/*
 * Forward-only counterpart (the inference side, render.py:169-171, :137-139, :294-296): the rendered feature map (C,H,W)
 * is resized (bilinear, align_corners=True) to (Hg,Wg) and - if weight / bias are given - decoded by the 1x1 conv into
 * `out` (Cout,Hg,Wg), fp32 or (out_is_half != 0) IEEE fp16 as render.py stores it.  Without a decoder Cout must equal C.
 * With a decoder C must be 32, 64 or 128.  `scratch`: f3dgs_feature_decode_scratch_bytes(...) bytes (0 without a decoder).
 */
size_t f3dgs_feature_decode_scratch_bytes(int C, int Hg, int Wg, int has_decoder);
int f3dgs_feature_decode(int C, int H, int W, int Cout, int Hg, int Wg, const float* feature_map, const float* weight,
                         const float* bias, void* out, int out_is_half, void* scratch, void* stream);

/*
 * One torch.optim.Adam step (no weight decay, no amsgrad: the reference's configuration,
 * scene/gaussian_model.py:163-178) over one tensor of n floats, in place; `step` is the 1-based step count of that
 * tensor.  param / grad / exp_avg / exp_avg_sq are device pointers.
 */
int f3dgs_adam_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                    double beta2, double eps, int step, void* stream /* hipStream_t */);
/*
 * The same step restricted to the rows (of `width` floats) whose row_mask byte is non-zero: the other rows keep
 * parameter and moments untouched.  An EXTENSION (row f-4's "sparse-aware Adam using radii > 0"), not the reference's
 * optimizer: torch.optim.Adam also moves never-visible Gaussians by their decaying first moment.  row_mask == NULL is
 * f3dgs_adam_step.
 */
int f3dgs_adam_step_rows(size_t n, size_t width, const uint8_t* row_mask, float* param, const float* grad, float* exp_avg,
                         float* exp_avg_sq, double lr, double beta1, double beta2, double eps, int step,
                         void* stream /* hipStream_t */);

/*
 * The same update over SEVERAL tensors in ONE launch (the reference steps the seven per-Gaussian tensors with torch's
 * foreach kernels, scene/gaussian_model.py:163-178): a table of up to F3DGS_ADAM_MAX_TENSORS entries, each with its own
 * learning rate and step count (torch keeps them per parameter); beta1 / beta2 / eps are shared.  `row_mask` (optional,
 * `rows` bytes): the visibility-masked variant of f3dgs_adam_step_rows, applied to every tensor whose `n` is a multiple of `rows`.
 */
#define F3DGS_ADAM_MAX_TENSORS 16
typedef struct {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    size_t n;        /* elements */
    double lr;
    int step;        /* counts from 1 */
} f3dgs_adam_tensor;
int f3dgs_adam_step_multi(int n_tensors, const f3dgs_adam_tensor* tensors /* host array */, double beta1, double beta2, double eps,
                          const uint8_t* row_mask, size_t rows, void* stream);

/*
 * Row movement of one densification (scene/gaussian_model.py:300-431: densify_and_clone, densify_and_split,
 * prune_points and the optimizer-state edits of cat_tensors_to_optimizer / _prune_optimizer) as ONE gather over all
 * per-Gaussian tensors.  Output row j of every tensor comes from source row src_row[j]; kind[j] says how:
 *   0 kept original (row and Adam moments copied), 1 clone (row copied, moments zero),
 *   2 split child (as a clone, but tensors in mode OVERRIDE_CHILD take row override_row[j] of `override_src`:
 *     the child's new xyz and scaling).
 * All pointers are device pointers; `dst` buffers are the caller's (capacity-sized, never the same memory as `src`).
 */
#define F3DGS_DENSIFY_COPY 0            /* parameters: always the source row */
#define F3DGS_DENSIFY_ZERO_NEW 1        /* optimizer moments: source row for kind 0, zeros for kinds 1 and 2 */
#define F3DGS_DENSIFY_OVERRIDE_CHILD 2  /* xyz, scaling: source row for kinds 0/1, override row for kind 2 */
#define F3DGS_DENSIFY_MAX_TENSORS 32
typedef struct f3dgs_densify_tensor {
    const float* src;           /* (n_in, width) */
    float* dst;                 /* (>= n_out, width) */
    const float* override_src;  /* (n_children, width) or NULL */
    int width;                  /* floats per row */
    int mode;                   /* F3DGS_DENSIFY_* */
} f3dgs_densify_tensor;
int f3dgs_densify_gather(size_t n_out, const int32_t* src_row, const uint8_t* kind, const int32_t* override_row, int n_tensors,
                         const f3dgs_densify_tensor* tensors /* host array */, void* stream /* hipStream_t */);

/*
 * Replaces SimpleKNN::knn / distCUDA2 of the reference's second native module (submodules/simple-knn/
 * simple_knn.cu:45-221, spatial.cu:15-25; caller scene/gaussian_model.py:146): mean_dist2[i] = mean of the squared
 * distances from point i to its three nearest neighbours (exact; a missing neighbour counts as FLT_MAX, as there).
 * points (P,3), mean_dist2 (P) device pointers; `scratch` = f3dgs_knn_scratch_bytes(P) bytes of device memory.
 * Everything is enqueued on `stream`; nothing is read back to the host (the reference synchronises twice).
 */
size_t f3dgs_knn_scratch_bytes(int P);
int f3dgs_knn_mean_dist2(int P, const float* points, float* mean_dist2, void* scratch, void* stream /* hipStream_t */);

/*
 * ---- Side channels of f3dgs_backward: threading contract -----------------------------------------------------------
 * The four setters below and f3dgs_set_feature_grad_lowres change how the f3dgs_backward calls OF THE CALLING HOST THREAD
 * behave.  All five are thread-local: they are invisible to, and unaffected by, any other thread.
 *   - the two callbacks and the accumulate switch are STICKY: they stay armed for every later call of the thread until reset
 *     (NULL / 0).  A caller arms them around exactly the calls it means (dp.py: a `with` block per backward pass);
 *   - f3dgs_set_feature_grad_lowres is ONE-SHOT: it is consumed - and cleared - by the next f3dgs_backward of the thread,
 *     whether that call succeeds or fails.
 * A second trainer in the same process therefore runs on its OWN thread (one thread per device is the supported shape:
 * the instance-count read-back words are kept per (thread, device, stream) as well) and arms its own channels there;
 * nothing needs a lock.  Two trainers that share a thread - interleaving their backward calls on it - must re-arm or
 * clear the channels between calls themselves: the library cannot tell the calls apart.  The process-wide OPTIONS
 * (f3dgs_set_option) are the only state shared between threads; they are read once at the top of each call.
 *
 * Optional notification inside f3dgs_backward (no counterpart in the reference): `fn(ctx, stream)` is called on
 * the calling host thread right after the blend backward has been ENQUEUED on `stream`, i.e. at the point of
 * the stream from which dL_dsemantic_feature is final while the per-Gaussian stage still follows.  A
 * data-parallel caller records an event there and starts the all-reduce of the feature gradient on another
 * stream, overlapping it with the rest of the call.  Thread-local; NULL removes it.
 */
typedef void (*f3dgs_stage_fn)(void* ctx, void* stream /* hipStream_t */);
void f3dgs_set_feature_grad_ready_callback(f3dgs_stage_fn fn, void* ctx);

/*
 * Gradient accumulation across views (no counterpart in the reference, which renders one view per step): with `on` != 0
 * the following f3dgs_backward calls of this host thread ADD their feature gradient into `dL_dsemantic_feature` instead
 * of overwriting it (the blend backward accumulates with atomics anyway: the zero-fill in front of it is skipped).  A
 * caller that renders several views per optimiser step hands in ONE (P, C) buffer, zeroed once, for all of them - no
 * per-view gradient tensor, no per-view zero-fill, no per-view add - and, data parallel, reduces that buffer from inside
 * the LAST view's backward pass (f3dgs_set_feature_grad_ready_callback).  Every other output is overwritten as always.
 * Thread-local; 0 restores the default.
 */
void f3dgs_set_feature_grad_accumulate(int on);

/*
 * Second optional notification inside f3dgs_backward (no counterpart in the reference): with a callback registered the
 * per-Gaussian stage (K8 + K9, R/cuda_rasterizer/backward.cu:145-404 - row-parallel) runs as `chunks` launches over
 * consecutive row ranges, and `fn(ctx, stream, row_begin, row_end)` is called on the calling host thread right after the
 * launch covering Gaussians [row_begin, row_end) has been ENQUEUED: from that point of the stream on, rows
 * [row_begin, row_end) of EVERY per-Gaussian output (dL_dmean3D, dL_dsh, dL_dscale, dL_drot, dL_dopacity, dL_dcolor,
 * dL_dmean2D, dL_dcov3D) are final.  A data-parallel caller starts the all-reduce of those rows (the SH gradient is
 * 48 of the 59 non-feature floats per Gaussian) while the later chunks still run.  Ranges are disjoint, ascending and
 * cover [0, P); row_begin is a multiple of 64.  chunks <= 1: one launch, one call.  Thread-local; NULL removes it.
 */
typedef void (*f3dgs_rows_fn)(void* ctx, void* stream /* hipStream_t */, int row_begin, int row_end);
void f3dgs_set_grad_rows_ready_callback(f3dgs_rows_fn fn, void* ctx, int chunks);

/*
 * Test / profiling hooks (not part of the reference surface).  They expose
 * the private state written by f3dgs_forward so that every stage can be
 * compared with the oracle in isolation.  Each copies `count` elements
 * starting at element 0 into a HOST buffer and synchronises the stream.
 *   what: "rec" (12 floats per Gaussian: mean_x, mean_y, conic a,b,c, opacity, r,g,b, depth, radius bits, pad;
 *         defined only where radii > 0)  "clamped"(u8 bitmask)  "tiles_touched"(u32)  "depth_key"(u32)
 *         "order"(u32 x P, depth order)  "counters"(16 x u32: [0] = entries of point_list,
 *         [1] = reference-style num_rendered)  "point_list"(u32 x R)  "tile_sorted"(u32 x R)
 *         "ranges"(uint2 per tile)  "final_T"(float per pixel)  "n_contrib"(u32 per pixel)
 */
int f3dgs_debug_read(
    const char* what,
    int P, int C, int R, int width, int height,
    const char* geom_buffer,
    const char* binning_buffer,
    const char* image_buffer,
    void* host_dst, size_t dst_bytes,
    void* stream);

/* Per-stage device-time accounting, active while option "profile" is 1 (seeded from F3DGS_PROFILE).  HIP events are recorded on the call's stream around every stage (no
 * synchronisation inside forward/backward); f3dgs_profile_read waits for the recorded events and
 * returns, per stage name, the accumulated milliseconds and the number of spans since the last
 * f3dgs_profile_reset.  Returns the number of stages written; names[i] are static strings. */
int f3dgs_profile_read(const char** names, double* total_ms, long* calls, int max_stages);
void f3dgs_profile_reset(void);

#ifdef __cplusplus
}
#endif

#endif /* F3DGS_H_INCLUDED */
