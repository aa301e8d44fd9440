// common.h — private declarations shared by the HIP translation units of libf3dgs_hip.so.
//
// gfx950 / CDNA4 only: wave64, 256 CUs in 8 XCDs, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <algorithm>
#include <stddef.h>
#include <stdint.h>

#include "../../include/f3dgs.h"

namespace f3dgs {

constexpr int TILE = 16;             // binning tile edge (reference: config.h:18-19)
constexpr int WAVE = 64;
constexpr size_t ALIGN = 256;        // alignment of every carved sub-buffer

// One packed splat record per Gaussian: everything the blend kernels need, gathered with three
// 16-byte loads.  (reference keeps means2D / conic_opacity / rgb / depths in four arrays.)
//   q0 = {mean_x, mean_y, conic_a, conic_b}   q1 = {conic_c, opacity, red, green}
//   q2 = {blue, depth, radius (int bits), unused}
struct SplatRec {
    float4 q0, q1, q2;
};

// Per-Gaussian gradient record accumulated by the blend backward (one 48-byte line):
//   [0] dL/dmean2D.x  [1] dL/dmean2D.y  [2..4] dL/dconic (a, b, c)  [5] dL/dopacity
//   [6..8] dL/dcolor  [9] dL/dz  [10..11] pad
constexpr int GREC = 12;

// ---- carving of the three opaque state buffers ---------------------------------------------------
struct Carver {
    char* base;
    size_t off;
    explicit Carver(char* b) : base(b), off(0) {}
    template <typename T>
    T* take(size_t count) {
        off = (off + ALIGN - 1) & ~(ALIGN - 1);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
    size_t total() const { return (off + ALIGN - 1) & ~(ALIGN - 1); }
};

constexpr int RADIX_BITS = 8;                 // tile-id passes
constexpr int RADIX_BINS = 1 << RADIX_BITS;
constexpr int DEPTH_RADIX_BITS = 11;          // depth-key passes: 11 + 11 + 10 bits
constexpr int DEPTH_RADIX_BINS = 1 << DEPTH_RADIX_BITS;
constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 8;                                   // fewest keys per thread of any pass (sizes the histograms)
constexpr int SORT_CHUNK = SORT_THREADS * SORT_ITEMS;           // items per workgroup
constexpr int SCAN_CHUNK = 256 * 16;
constexpr int SMALL_SORT_MAX = 16384;                           // pairs the one-launch LDS-resident sort takes (binning.hip: small_sort_kernel)
constexpr int BIGQ_CAP = 1024;                                  // slots of the big-splat queue of the emit kernel (4 x 256: zeroed by four workgroups)
constexpr int DEPTH_SORT_ITEMS = 8;                             // onesweep depth passes: 2048 keys per workgroup

inline size_t sort_blocks(size_t n) { return (n + SORT_CHUNK - 1) / SORT_CHUNK; }
inline size_t scan_blocks(size_t n) { return (n + SCAN_CHUNK - 1) / SCAN_CHUNK; }
inline size_t depth_sort_blocks(size_t n) { return (n + SORT_THREADS * DEPTH_SORT_ITEMS - 1) / (SORT_THREADS * DEPTH_SORT_ITEMS); }
inline size_t emit_blocks(size_t n) { return (n + 255) / 256; }

struct GeomState {
    SplatRec* rec;            // P
    uint8_t* clamped;         // P (bit0..2 = r,g,b clamped)
    uint32_t* tiles_touched;  // P
    uint32_t* depth_key;      // P   float bits of view-space depth, 0xFFFFFFFF when culled
    uint32_t* key_a;          // P   ping-pong buffers of the depth sort
    uint32_t* key_b;
    uint32_t* val_a;
    uint32_t* val_b;
    uint32_t* hist;           // RADIX_BINS * sort_blocks(P) + RADIX_BINS
    uint32_t* scan_tmp;       // scan_blocks(P) + 8      list entries of every chunk of SCAN_CHUNK Gaussians in depth order
    uint32_t* scan_sub;       // 64 x scan_blocks(P)     list entries of every run of 64 Gaussians in depth order
    uint32_t* ref_partial;    // per preprocess workgroup: bounding-rectangle tile counts, then list-entry counts, then 1 if a visible Gaussian of the workgroup has a long axis
    uint32_t* counters;       // 16 words: [0] = instances in the (culled) lists, [1] = reference num_rendered, [2] != 0: a visible Gaussian has a long axis, [3] = entries the binning buffer was carved for (0 until the emit kernel ran)
    // -- control words of the single-pass sorts (lookback.h).  depth_hist is zeroed by preprocess_kernel (it is
    //    accumulated by the kernel after it); everything from lb_words on is zeroed by sort_prologue_kernel.
    uint32_t* depth_hist;     // 4 x 256 digit totals of the depth keys
    // -- queue of the splats of more than EMIT_BIG tiles (emit kernel, work stealing): big_ctl[0] slots handed out, [1] the
    //    stealers' cursor, [4 + i] state of slot i (0 empty, 1 published, 2 claimed); zeroed by preprocess_kernel
    uint32_t* big_ctl;        // 4 + BIGQ_CAP
    uint2* big_items;         // BIGQ_CAP x {Gaussian, first list position}
    uint32_t* lb_words;       // start of the zero-filled region, n_lb_words long:
    size_t n_lb_words;
    uint32_t* tickets;        //   8  arrival counters: [0..3] depth passes, [4] emit, [5..6] tile passes, [7] emit "done"
    uint32_t* tile_hist;      //   2 x 256 digit totals of the tile ids (written by the last emit workgroup)
    uint32_t* depth_status;   //   4 x depth_sort_blocks(P) x 256
    uint32_t* emit_status;    //   emit_blocks(P)
    uint32_t* hist_copies;    //   64 x 512 private copies of the tile digit histograms (emit kernel)
    static GeomState carve(char* base, size_t P, size_t* bytes) {
        Carver c(base);
        GeomState g;
        g.rec = c.take<SplatRec>(P);
        g.clamped = c.take<uint8_t>(P);
        g.tiles_touched = c.take<uint32_t>(P);
        g.depth_key = c.take<uint32_t>(P);
        g.key_a = c.take<uint32_t>(P);
        g.key_b = c.take<uint32_t>(P);
        g.val_a = c.take<uint32_t>(P);
        g.val_b = c.take<uint32_t>(P);
        g.hist = c.take<uint32_t>(DEPTH_RADIX_BINS * sort_blocks(P) + DEPTH_RADIX_BINS);
        g.scan_tmp = c.take<uint32_t>(scan_blocks(P) + 8);
        g.scan_sub = c.take<uint32_t>(64 * scan_blocks(P));
        g.ref_partial = c.take<uint32_t>(3 * ((P + 255) / 256) + 3);
        g.counters = c.take<uint32_t>(16);
        g.depth_hist = c.take<uint32_t>(4 * 256);
        g.big_ctl = c.take<uint32_t>(4 + BIGQ_CAP);
        g.big_items = c.take<uint2>(BIGQ_CAP);
        g.n_lb_words = 8 + 2 * 256 + 4 * depth_sort_blocks(P) * 256 + emit_blocks(P) + 64 * 512;
        g.lb_words = c.take<uint32_t>(g.n_lb_words);
        g.tickets = g.lb_words;
        g.tile_hist = g.tickets + 8;
        g.depth_status = g.tile_hist + 2 * 256;
        g.emit_status = g.depth_status + 4 * depth_sort_blocks(P) * 256;
        g.hist_copies = g.emit_status + emit_blocks(P);
        if (bytes) *bytes = c.total();
        return g;
    }
};

struct BinState {
    uint32_t* point_list;   // N  (final, sorted by tile then depth)
    uint32_t* tile_sorted;  // N  tile id of every entry of point_list
    uint32_t* id_tmp;       // N  ping-pong partners
    uint32_t* tile_tmp;     // N
    uint32_t* hist;         // RADIX_BINS * sort_blocks(N) + RADIX_BINS
    uint32_t* tile_status;  // 2 x sort_blocks(N) x 256 look-back words (zeroed by the emit kernel)
    uint2* ranges_enc;      // tiles: {min position, UINT_MAX - (max position + 1)} per tile, preset to all-ones and
                            // updated with atomicMin by the final tile-sort pass; decoded into ImageState::ranges
                            // by the blend forward kernel (an untouched entry = empty tile)
    // The sorted list is the FIRST sub-buffer: its address does not depend on N, so a reader that only knows the
    // buffer (the backward pass) finds it without the length the forward carved with.
    static const uint32_t* list_of(const char* base) { return reinterpret_cast<const uint32_t*>(base); }
    static BinState carve(char* base, size_t N, size_t* bytes, size_t tiles = 0) {
        Carver c(base);
        BinState b;
        b.point_list = c.take<uint32_t>(N);   // offset 0 (see list_of)
        b.tile_sorted = c.take<uint32_t>(N);
        b.id_tmp = c.take<uint32_t>(N);
        b.tile_tmp = c.take<uint32_t>(N);
        b.hist = c.take<uint32_t>(RADIX_BINS * sort_blocks(N) + RADIX_BINS);
        b.tile_status = c.take<uint32_t>(2 * sort_blocks(N) * 256);
        b.ranges_enc = c.take<uint2>(tiles);
        if (bytes) *bytes = c.total();
        return b;
    }
};

struct ImageState {
    float* final_T;        // H*W
    uint32_t* n_contrib;   // H*W
    uint2* ranges;         // tiles
    uint32_t* tile_len;    // tiles   longest walk of the tile = max n_contrib of its pixels (blend forward, atomicMax)
    uint32_t* tile_order;  // tiles   tiles by descending tile_len: workgroup order of the pixel-lane blend backward
    static ImageState carve(char* base, size_t HW, size_t tiles, size_t* bytes) {
        Carver c(base);
        ImageState s;
        s.final_T = c.take<float>(HW);
        s.n_contrib = c.take<uint32_t>(HW);
        s.ranges = c.take<uint2>(tiles);
        s.tile_len = c.take<uint32_t>(tiles);
        s.tile_order = c.take<uint32_t>(tiles);
        if (bytes) *bytes = c.total();
        return s;
    }
};

// ---- process-wide options ------------------------------------------------------------------------
// Seeded ONCE from the environment (F3DGS_<NAME>) when the library is first used and changed afterwards only
// through f3dgs_set_option(); no launch path ever calls getenv().  They select between complete, tested code
// paths (never skip work): see include/f3dgs.h for the list.
struct Options {
    int tile_cull;       // 1: drop instances whose 1/255 ellipse misses the tile (default); 0: reference-identical lists
    int feature_mfma;    // 1: feature contraction on the matrix pipe where a kernel variant exists (default)
    int profile;         // 1: record per-stage HIP events (f3dgs_profile_read)
    int bwd_half;        // instance-lane blend backward: chunks of 32 instances against two pixel halves (default 1); 0: 64-lane chunks
    int bwd_order;       // blend backward: workgroups take the tiles longest walk first (default 1)
    int bwd_m44;         // pixel-lane blend backward: the colour / depth sums on 4 x 4 matrix blocks (default 1)
    int bwd_split16;     // pixel-lane blend backward, up to 16 channels: feature and moment blocks split over the waves by quadrants (default 1)
    int bwd_bf16;        // pixel-lane blend backward: every contraction on bf16 matrix instructions, operands as two bf16 terms: 1 always, 0 never (exact fp32), -1 (default) while no visible Gaussian's axis ratio exceeds bwd_bf16_max_ratio
    int bwd_bf16_max_ratio;   // (default 16)
    int bwd_wide8;       // pixel-lane blend backward, bf16 shape: later channel windows of up to 128 channels on eight waves per tile where more than 64 channels remain (default 1)
    int bwd_pl;          // blend backward: pixel-lane formulation with all sums on the matrix pipe: 1 always, 0 never, -1 (default) for C > 0 (C > 4 with bwd_bf16 = 0); needs feature_mfma
    int fwd_wide;        // blend forward: 128-channel windows where more than 64 channels remain (default 1)
    int fwd_solo;        // blend forward, one quadrant per wave: one 64-thread workgroup per quadrant (default 1; the waves never synchronise)
    int sort_onesweep;   // 1: single-pass radix passes with decoupled look-back (measured slower on MI355X; default 0)
    int sync_free;       // -1 (default): as 1 inside a graph capture, as 0 otherwise; 1: the forward call never waits for the instance count in the middle of its enqueue - binning buffers of a CAPACITY, kernels that read the count on the device, the count checked behind the last launch (one retry) - and can be captured in a HIP graph; 0: the blocking read, capture refused
    int instance_capacity;   // sync_free: entries of the instance lists to provide for (0, default: 1.25 x the thread's last count on the device + 4096)
#ifdef F3DGS_DEV
    int dev;             // development builds only (make DEV=1): work-skipping experiments, never in a release library
#endif
};
Options& options();

// ---- host-side launchers (one per translation unit) ----------------------------------------------
// Camera matrices / position / background stay in DEVICE memory (that is where the reference's tensors
// live); kernels read them through uniform (scalar) loads, so no call ever synchronises to fetch them.
struct ViewParams {
    const float* view;    // 16 floats, element (r, c) at [4c + r]
    const float* proj;    // 16 floats
    const float* campos;  // 3 floats (may be null when colours are precomputed)
    const float* bg;      // 3 floats
    float tanx, tany, fx, fy;
    int W, H, gx, gy;
    int band0, band1;       // forward only: tile rows [band0, band1) this call lists and blends (0, gy: the whole view; f3dgs_set_tile_band)
    float scale_modifier;
    float max_axis_ratio;   // forward only: preprocess reports whether a visible Gaussian is longer than this times its width
};

// preprocess.hip (built with -ffp-contract=off)
void launch_mark_visible(int P, const float* means3D, const float* view_dev, uint8_t* present, hipStream_t s);
void launch_preprocess(int P, int D, int M, const float* means3D, const float* scales, const float* rotations,
                       const float* opacities, const float* shs, const float* cov3D_precomp,
                       const float* colors_precomp, const ViewParams& vp, int* radii, GeomState g, int cull, hipStream_t s);
void launch_preprocess_backward(int P, int D, int M, int C, const float* means3D, const int* radii, const float* shs,
                                const float* scales, const float* rotations, const float* cov3D_precomp,
                                const ViewParams& vp, const GeomState& g, const float* grec, float* dL_dmean2D,
                                float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                                float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dz,
                                int row_begin, int row_end, hipStream_t s);
int preprocess_backward_row_align();     // row_begin of a partial launch must be a multiple of this

// binning.hip
// Two-level sums of in[gather[i]] (or in[i]) for the emit kernel: chunk_sums[c] = total of chunk c (SCAN_CHUNK items),
// sub[64 c + r] = sum of run r (64 items) of chunk c.
void launch_offset_sums(const uint32_t* in, const uint32_t* gather, size_t n, uint32_t* chunk_sums, uint32_t* sub,
                        hipStream_t s);
// Stable LSD radix sort of (key,val) u32 pairs on key bits [0, nbits).  Result lands in (key_out,val_out);
// (key_in,val_in) and the *_tmp buffers are clobbered.  key_out/val_out may alias the tmp or in buffers
// only as arranged by the caller through the pass parity (see binning.hip).
// `ranges_enc` (optional): the final pass records every tile's [min, max + 1) output positions (BinState::ranges_enc)
// `n_dev` (optional; the sync-free forward): n is the capacity the buffers were carved for and sizes every launch, the kernels
// take the item count from this device word (clamped to n)
void launch_radix_sort_pairs(uint32_t* key_a, uint32_t* val_a, uint32_t* key_b, uint32_t* val_b, size_t n, int nbits,
                             uint32_t* hist, bool result_in_a, uint2* ranges_enc, hipStream_t s, const uint32_t* n_dev = nullptr);
// also presets `ranges_enc` (all-ones = "no entry yet") for the final tile-sort pass and zeroes `tile_len`
// `cap`: entries the instance arrays hold.  `n_dev` (null: cap IS the list length): the device's count - a frame whose count
// exceeds cap stores nothing and raises `overflow` (device-visible host word, may be null).  The capacity is left in
// GeomState::counters[3] for readers of the binning buffer.
void launch_emit_instances(int P, const GeomState& g, const uint32_t* order, int gx, int gy, int2 band, int cull,
                           uint32_t* inst_tile, uint32_t* inst_id, uint2* ranges_enc, uint32_t* tile_len, uint32_t cap,
                           const uint32_t* n_dev, uint32_t* overflow, hipStream_t s);
// single-pass flavour (option sort_onesweep): offsets by decoupled look-back inside the emit kernel, which also
// produces the tile digit histograms, presets `ranges` for the final sort pass and zero-fills b.tile_status
void launch_emit_scan(int P, const GeomState& g, const BinState& b, const uint32_t* order, int gx, int gy, int2 band, int cull,
                      uint32_t* inst_tile, uint32_t* inst_id, uint32_t N, uint2* ranges, hipStream_t s);
void launch_sort_prologue(const GeomState& g, size_t P, hipStream_t s);
void launch_depth_sort_onesweep(const GeomState& g, size_t n, hipStream_t s);
void launch_tile_sort_onesweep(const GeomState& g, const BinState& b, size_t n, int passes, uint2* ranges, hipStream_t s);
// `tj` (optional): the first kernel of the sort also produces the two instance totals from the per-workgroup partials of
// preprocess_kernel - device counters and the host's pinned words - and `ready` is recorded right behind it.
// The two-level list-offset sums of launch_offset_sums, produced by the depth sort's own launch where that is ONE workgroup
// (up to SMALL_SORT_MAX Gaussians): a launch less per frame (c1: ~9 us of ~115 replayed).  All null: not wanted.
struct OffsetSumsJob {
    const uint32_t* tiles_touched;
    uint32_t* chunk_sums;
    uint32_t* sub;
};
struct TotalsJob {
    const uint32_t* partial;   // GeomState::ref_partial
    int n_partial;
    uint32_t* counters;        // GeomState::counters
    uint32_t* host;            // pinned, device-visible (may be null)
    hipEvent_t ready;          // host side only
};
// `sums` (optional) / `sums_done`: see OffsetSumsJob - *sums_done says whether this call produced them
hipError_t launch_depth_sort(const uint32_t* keys, uint32_t* key_a, uint32_t* val_a, uint32_t* key_b, uint32_t* val_b, size_t n,
                             uint32_t* hist, const TotalsJob* tj, hipStream_t s, const OffsetSumsJob* sums = nullptr,
                             bool* sums_done = nullptr);
void launch_iota(uint32_t* dst, size_t n, hipStream_t s);
void launch_radix_sort_keys_to_order(const uint32_t* keys, uint32_t* key_a, uint32_t* val_a, uint32_t* key_b, uint32_t* val_b,
                                     size_t n, int nbits, uint32_t* hist, hipStream_t s);

// feature_loss.hip
size_t feature_l1_scratch_bytes(int C, int Cout, int Hg, int Wg, bool decoder);
bool feature_l1_decoder_supported(int C);
const float* feature_l1_lowres_grad(char* scratch, int C, int Cout, int Hg, int Wg, bool decoder);   // (Hg Wg, C) pixel-major
hipError_t launch_feature_l1(int C, int H, int W, int Cout, int Hg, int Wg, const float* feature_map, const float* weight,
                             const float* bias, const float* gt, float* loss, float* d_feature_map, float* d_weight,
                             float* d_bias, char* scratch, hipStream_t s);

size_t feature_decode_scratch_bytes(int C, int Hg, int Wg, bool decoder);
hipError_t launch_feature_decode(int C, int H, int W, int Cout, int Hg, int Wg, const float* feature_map, const float* weight,
                                 const float* bias, void* out, bool half, char* scratch, hipStream_t s);

// adam.hip
void launch_adam_step(size_t n, float* p, const float* g, float* m, float* v, double lr, double b1, double b2, double eps,
                      int step, const uint8_t* row_mask, size_t width, hipStream_t s);

void launch_adam_step_multi(int count, const f3dgs_adam_tensor* tensors, double b1, double b2, double eps, const uint8_t* row_mask,
                            size_t rows, hipStream_t s);

// densify.hip
using DensifyTensor = f3dgs_densify_tensor;
constexpr int DENSIFY_MAX_TENSORS = F3DGS_DENSIFY_MAX_TENSORS;
constexpr int DENSIFY_ZERO_NEW = F3DGS_DENSIFY_ZERO_NEW, DENSIFY_OVERRIDE_CHILD = F3DGS_DENSIFY_OVERRIDE_CHILD;
void launch_densify_gather(size_t n_out, const int32_t* src_row, const uint8_t* kind, const int32_t* override_row, int n_tensors,
                           const DensifyTensor* tensors, hipStream_t s);

// knn.hip
size_t knn_scratch_bytes(size_t P);
void launch_knn_mean_dist2(int P, const float* points, float* out, char* scratch, hipStream_t s);

// render_fwd.hip / render_bwd.hip
// `tile_len` (zero-filled by the caller): receives the longest walk of every tile (max n_contrib of its pixels)
void launch_render_forward(const ViewParams& vp, int C, const uint2* ranges_enc, uint2* ranges, const uint32_t* point_list,
                           const SplatRec* rec, const float* feat, float* final_T,
                           uint32_t* n_contrib, float* out_color, float* out_feat, float* out_depth, uint32_t* tile_len,
                           hipStream_t s);      // (vp.band0 / band1: see band_perm)
// The feature-map gradient at the resolution of the loss (f3dgs_set_feature_grad_lowres): (Hg Wg, C) pixel-major; the blend
// backward applies the transposed bilinear resize while it stages a tile.  `scale`: device scalar multiplied in, or null.
struct LowresGrad {
    const float* gx = nullptr;
    const float* scale = nullptr;
    int Hg = 0, Wg = 0;
};
// `tile_len` / `tile_order`: the pixel-lane kernel takes its tiles longest walk first (order built here, one small launch)
// `dL_dfeat` may be null when `lowres` carries the feature-map gradient (given both, the kernel adds them)
// `contraction` of the pixel-lane kernel (api.hip decides; see option bwd_bf16): 1 bf16 matrix instructions with two-term operands,
// 2 the hybrid shape (first window: bf16 feature / colour blocks, the moment block in exact fp32), 0 exact fp32, 3 (a frame the
// host never read: inside a graph) 1 or 2 by the device word `gate` (GeomState::counters + 2: != 0 = a visible Gaussian has a
// long axis) - both first-window kernels are launched, the workgroups of one leave at once.
// Returns what ran: 1 / 2 / 3 the pixel-lane kernel in that shape, 0 anything exact (its fp32 shape, the instance-lane kernel).
int launch_render_backward(const ViewParams& vp, int C, const uint2* ranges, const uint32_t* point_list,
                           const SplatRec* rec, const float* final_T, const uint32_t* n_contrib,
                           const float* dL_dpix, const float* dL_dfeat, const float* dL_ddepth, float* grec,
                           float* dL_dfeature, const uint32_t* tile_len, uint32_t* tile_order, const LowresGrad* lowres,
                           int contraction, const uint32_t* gate, hipStream_t s);      // (vp.band0 / band1: the forward call's band, see band_perm)
struct BwdArgs;
void launch_render_backward_pl(BwdArgs a, int C, hipStream_t s);     // render_bwd_pl.hip
void launch_tile_order(const uint32_t* tile_len, size_t tiles, uint32_t* order, uint32_t band_b0, uint32_t band_tb, hipStream_t s);

// ---- device helpers --------------------------------------------------------------------------------
#if defined(__HIPCC__)
// XCD-aware bijective remap of a linear workgroup id: workgroup b runs on XCD b % 8 (observed,
// MI355X_MICROARCH.md), so give every XCD one contiguous run of tiles to keep neighbouring tiles
// (which share splat records and feature vectors) behind the same L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    constexpr uint32_t X = 8;
    const uint32_t q = n / X, r = n % X;
    const uint32_t xcd = b % X, k = b / X;
    const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}
// A view of which only a band of tile rows is listed (f3dgs_set_tile_band): the band is a contiguous eighth (or less) of the
// tile ids, i.e. ONE XCD's run under xcd_remap - seven XCDs would blend nothing.  band_perm relabels the tiles so that every
// XCD's run of VIRTUAL ids holds a contiguous eighth of the band's tiles followed by an eighth of the others: a bijection of
// [0, T); tb = 0: identity.  (b0 = first tile of the band, tb = tiles in it; the caller guarantees T / 8 - tb / 8 >= 8.)
__host__ __device__ __forceinline__ uint32_t band_perm(uint32_t v, uint32_t T, uint32_t b0, uint32_t tb) {
    if (tb == 0) return v;
    constexpr uint32_t X = 8;
    const uint32_t q = T / X, r = T % X;
    const uint32_t x = v < r * (q + 1) ? v / (q + 1) : r + (v - r * (q + 1)) / q;
    const uint32_t f = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    const uint32_t k = v - f;
    const uint32_t qb = tb / X, rb = tb % X;
    const uint32_t bf = x < rb ? x * (qb + 1) : rb * (qb + 1) + (x - rb) * qb;
    const uint32_t nb = x < rb ? qb + 1 : qb;
    if (k < nb) return b0 + bf + k;
    const uint32_t j = (f - bf) + (k - nb);      // index among the tiles outside the band
    return j < b0 ? j : j + tb;
}
#endif
// host side: (first tile, tiles) of the band for band_perm, or (0, 0) where the relabelling is off (whole view, a band of
// more than half the grid, a grid too small to matter)
inline void band_perm_params(int gx, int gy, int band0, int band1, uint32_t* b0, uint32_t* tb) {
    *b0 = 0; *tb = 0;
    if (band0 <= 0 && band1 >= gy) return;
    const uint32_t T = (uint32_t)gx * (uint32_t)gy, n = (uint32_t)gx * (uint32_t)std::max(0, band1 - band0);
    if (n == 0 || T / 8 < n / 8 + 8) return;
    *b0 = (uint32_t)gx * (uint32_t)band0; *tb = n;
}

}  // namespace f3dgs
