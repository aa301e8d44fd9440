// api.hip — extern "C" entry points of libf3dgs_hip.so (declared in include/f3dgs.h).
//
// Host orchestration of the stages.  Replaces CudaRasterizer::Rasterizer::{markVisible,forward,backward}
// (reference: submodules/diff-gaussian-rasterization-feature/cuda_rasterizer/rasterizer_impl.cu:141-153,
// 198-342, 347-461).  Everything is enqueued on the caller's stream; the only host synchronisation is
// the 4-byte read-back of num_rendered (same place as rasterizer_impl.cu:283).

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

using namespace f3dgs;

namespace {

thread_local std::string g_err;
thread_local f3dgs_stage_fn g_feature_ready_fn = nullptr;
thread_local void* g_feature_ready_ctx = nullptr;
thread_local f3dgs_rows_fn g_rows_ready_fn = nullptr;
thread_local void* g_rows_ready_ctx = nullptr;
thread_local int g_rows_ready_chunks = 1;
thread_local int g_feature_accumulate = 0;
thread_local LowresGrad g_lowres;      // consumed by the next f3dgs_backward of this thread
thread_local int g_band_begin = 0, g_band_end = 0;     // f3dgs_set_tile_band: tile rows the forward calls of this thread list (0, 0: all)
std::atomic<int> g_last_bwd_bf16{-1};  // contraction of the process's last blend backward (PyTorch runs it on an autograd thread): 1 bf16 two-term, 0 exact fp32, -1 none yet

// What f3dgs_forward learned about a frame that the matching f3dgs_backward needs on the HOST: the largest axis ratio among
// the frame's visible Gaussians (option bwd_bf16 = -1 chooses the contraction precision by it).  The two calls share nothing
// but the caller's state buffers - and may run on different host threads (PyTorch's autograd engine runs the backward pass on a
// thread of its own) - so the value is kept here, process-wide, keyed by (device, geometry buffer): the caller hands that buffer
// back untouched, and a buffer that has been re-used by a later forward call belongs to that call.  A backward call that finds
// nothing (a buffer copied elsewhere, more than 256 frames in flight) takes the exact contraction.
struct FrameNote { int device; const void* geom; float axis_ratio; int band0, band1; };      // band: the tile rows the forward call listed (f3dgs_set_tile_band)      // axis_ratio: 0 no visible Gaussian with a long axis, 1 some, 2 ask the device (a captured frame)
std::mutex g_frames_mu;
std::vector<FrameNote> g_frames;
void note_frame(const void* geom, float axis_ratio, int band0, int band1) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_frames_mu);
    for (FrameNote& f : g_frames)
        if (f.device == dev && f.geom == geom) { f.axis_ratio = axis_ratio; f.band0 = band0; f.band1 = band1; return; }
    if (g_frames.size() >= 256) g_frames.erase(g_frames.begin());
    g_frames.push_back({dev, geom, axis_ratio, band0, band1});
}
bool frame_axis_ratio(const void* geom, float* axis_ratio, int* band0 = nullptr, int* band1 = nullptr) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_frames_mu);
    for (const FrameNote& f : g_frames)
        if (f.device == dev && f.geom == geom) {
            *axis_ratio = f.axis_ratio;
            if (band0) *band0 = f.band0;
            if (band1) *band1 = f.band1;
            return true;
        }
    return false;
}

// Zero-fill inside a graph capture: hipMemsetAsync is captured as a memset node, and with the HIP runtime this library meets
// under PyTorch (ROCm 7.0) such nodes fill with the wrong pattern from the graph's SECOND replay on (measured: every row of
// dL_dsemantic_feature began with 16 bytes of a kernel-argument block; tools/graph_memset_probe.py).  A kernel node has no such
// problem, so a captured call clears its buffers with this kernel; eager calls keep hipMemsetAsync (6+ TB/s).
__global__ void __launch_bounds__(256) zero_fill_kernel(uint4* __restrict__ p, size_t n16, int tail_words) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < tail_words) reinterpret_cast<uint32_t*>(p + n16)[threadIdx.x] = 0u;
}
hipError_t zero_fill(void* p, size_t bytes, bool capturing, hipStream_t s) {
    if (!capturing || (reinterpret_cast<uintptr_t>(p) & 15) || (bytes & 3)) return hipMemsetAsync(p, 0, bytes, s);
    if (bytes == 0) return hipSuccess;
    const size_t n16 = bytes / 16;
    const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((n16 + 255) / 256, 256 * 16));
    hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, s, static_cast<uint4*>(p), n16, (int)((bytes & 15) / 4));
    return hipGetLastError();
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(F3DGS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---- optional per-stage profiling (F3DGS_PROFILE=1) ----------------------------------------------------
// HIP events are recorded on the caller's stream around every stage and resolved lazily when the
// totals are read, so the timed region itself is never synchronised.
bool profiling() { return options().profile != 0; }

// One event per stage boundary: the event that ends span i starts span i+1 (every recorded event costs a marker
// packet between two kernels, ~4 us of device time each).  Events are pooled; a span with a null name only hands
// its start event back to the pool (end of a chain).
struct PendingSpan {
    const char* name;
    hipEvent_t a, b;
};
std::mutex g_prof_mu;
std::vector<PendingSpan> g_pending;
std::vector<hipEvent_t> g_event_pool;
std::vector<std::pair<const char*, std::pair<double, long>>> g_totals;  // name -> (ms, calls)

void resolve_pending_locked();
constexpr size_t MAX_PENDING_SPANS = 4096;   // unread spans are folded into the totals beyond this

hipEvent_t take_event_locked() {
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// RAII: the chain is closed (its last event handed back) on every exit path of forward/backward.
struct StageTimer {
    hipStream_t s;
    bool on;
    bool blend_only;      // option profile = 2: only the two blend kernels are bracketed (4 events per step instead of 11)
    hipEvent_t prev;
    explicit StageTimer(hipStream_t st, bool allowed = true) : s(st), on(allowed && profiling()), blend_only(options().profile == 2), prev(nullptr) {
        if (on && !blend_only) {
            {
                std::lock_guard<std::mutex> lk(g_prof_mu);
                prev = take_event_locked();
            }
            (void)hipEventRecord(prev, s);
        }
    }
    ~StageTimer() {
        if (prev) {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            g_pending.push_back({nullptr, prev, nullptr});
        }
    }
    StageTimer(const StageTimer&) = delete;
    StageTimer& operator=(const StageTimer&) = delete;
    void mark(const char* name) {
        if (!on) return;
        if (blend_only) {
            const bool opens = strcmp(name, "tile_sort") == 0 || strcmp(name, "zero") == 0;      // the stage before a blend kernel
            const bool closes = strcmp(name, "render_fwd") == 0 || strcmp(name, "render_bwd") == 0;
            if (!opens && !closes) return;
            hipEvent_t ev;
            {
                std::lock_guard<std::mutex> lk(g_prof_mu);
                if (g_pending.size() >= MAX_PENDING_SPANS) resolve_pending_locked();
                ev = take_event_locked();
            }
            (void)hipEventRecord(ev, s);
            if (opens) { prev = ev; return; }           // (the destructor hands it back if no blend kernel follows)
            std::lock_guard<std::mutex> lk(g_prof_mu);
            if (prev) g_pending.push_back({name, prev, ev});
            g_pending.push_back({nullptr, ev, nullptr});
            prev = nullptr;
            return;
        }
        hipEvent_t e;
        {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            if (g_pending.size() >= MAX_PENDING_SPANS) resolve_pending_locked();   // nobody is reading: bound the queue
            e = take_event_locked();
        }
        (void)hipEventRecord(e, s);
        {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            g_pending.push_back({name, prev, e});
        }
        prev = e;
    }
};

void resolve_pending() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    resolve_pending_locked();
}

void resolve_pending_locked() {
    for (auto& p : g_pending) {
        if (p.name) {
            (void)hipEventSynchronize(p.b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, p.a, p.b);
            bool found = false;
            for (auto& t : g_totals)
                if (strcmp(t.first, p.name) == 0) { t.second.first += ms; t.second.second++; found = true; break; }
            if (!found) g_totals.push_back({p.name, {ms, 1}});
        } else {
            (void)hipEventSynchronize(p.a);
        }
        g_event_pool.push_back(p.a);     // every event is the start of exactly one span (the chain's last: of the sentinel)
    }
    g_pending.clear();
}

int check_debug(int debug, hipStream_t s, const char* stage) {
    if (!debug) return F3DGS_OK;
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail(F3DGS_ERR_HIP, "stage '%s' failed: %s", stage, hipGetErrorString(e));
    return F3DGS_OK;
}

void fill_view(ViewParams& vp, const float* view, const float* proj, const float* campos, const float* bg, float tanx,
               float tany, int W, int H, float mod) {
    vp.view = view; vp.proj = proj; vp.campos = campos; vp.bg = bg;
    vp.tanx = tanx; vp.tany = tany;
    vp.fx = W / (2.0f * tanx); vp.fy = H / (2.0f * tany);
    vp.W = W; vp.H = H;
    vp.gx = (W + TILE - 1) / TILE; vp.gy = (H + TILE - 1) / TILE;
    vp.band0 = 0; vp.band1 = vp.gy;
    vp.scale_modifier = mod;
    vp.max_axis_ratio = 0.f;
}

// option "tile_cull" = 0 keeps the reference's bounding-rectangle instance lists (bit-identical intermediate
// state, used by the parity tests); the default drops instances that cannot blend in a tile.
int tile_cull_enabled() { return options().tile_cull ? 1 : 0; }

struct OptionDesc {
    const char* name;
    const char* env;
    int Options::*field;
    int dflt;
};
const OptionDesc kOptions[] = {
    {"tile_cull", "F3DGS_TILE_CULL", &Options::tile_cull, 1},
    {"feature_mfma", "F3DGS_FEATURE_MFMA", &Options::feature_mfma, 1},
    {"profile", "F3DGS_PROFILE", &Options::profile, 0},
    {"bwd_half", "F3DGS_BWD_HALF", &Options::bwd_half, 1},
    {"bwd_pl", "F3DGS_BWD_PL", &Options::bwd_pl, -1},
    {"bwd_order", "F3DGS_BWD_ORDER", &Options::bwd_order, 1},
    {"bwd_m44", "F3DGS_BWD_M44", &Options::bwd_m44, 1},
    {"bwd_split16", "F3DGS_BWD_SPLIT16", &Options::bwd_split16, 1},
    {"bwd_bf16", "F3DGS_BWD_BF16", &Options::bwd_bf16, -1},
    {"bwd_bf16_max_ratio", "F3DGS_BWD_BF16_MAX_RATIO", &Options::bwd_bf16_max_ratio, 16},
    {"bwd_wide8", "F3DGS_BWD_WIDE8", &Options::bwd_wide8, 1},
    {"fwd_wide", "F3DGS_FWD_WIDE", &Options::fwd_wide, 1},
    {"fwd_solo", "F3DGS_FWD_SOLO", &Options::fwd_solo, 1},
    {"sort_onesweep", "F3DGS_SORT_ONESWEEP", &Options::sort_onesweep, 0},
    {"sync_free", "F3DGS_SYNC_FREE", &Options::sync_free, -1},
    {"instance_capacity", "F3DGS_INSTANCE_CAPACITY", &Options::instance_capacity, 0},
#ifdef F3DGS_DEV
    {"dev", "F3DGS_DEV_BITS", &Options::dev, 0},
#endif
};

// The per-Gaussian kernels read rotations, SH rows and feature rows - and write their gradients - with 16-byte accesses
// (include/f3dgs.h, "Alignment"): a base pointer off a 16-byte boundary is refused here instead of faulting in a kernel.
bool misaligned16(std::initializer_list<const void*> ptrs) {
    uintptr_t bits = 0;
    for (const void* p : ptrs) bits |= reinterpret_cast<uintptr_t>(p);
    return (bits & 15) != 0;
}

int tile_bits(int tiles) {
    int b = 0;
    while ((1 << b) < tiles) b++;
    return b < 1 ? 1 : b;
}

}  // namespace

namespace f3dgs {
Options& options() {
    static Options o = [] {
        Options v{};
        for (const OptionDesc& d : kOptions) {
            const char* e = getenv(d.env);   // read once, at first use of the library
            v.*(d.field) = e ? atoi(e) : d.dflt;
        }
        return v;
    }();
    return o;
}
}  // namespace f3dgs

namespace {
// Pinned landing zone + event for the instance-count read-back: one per (host thread, device, stream), created on first use
// and kept for the life of the thread - forward calls in flight on two streams of one thread (a multi-view step that overlaps
// the binning of view v + 1 with the blend kernels of view v, dp.py) never share a slot.  The words are allocated portable and
// mapped, and the kernel that stores the totals receives the DEVICE-side address of the mapping for the device it runs on.
struct CountReadback {
    uint32_t* host = nullptr;       // host address (read by this thread): [0] entries of our lists, [1] the reference's count, [2] long-axis flag (kernel-written);
                                    // [3] entries the last enqueued frame provided for (host-written); [4] sticky: an emit wave of a sync-free frame found no room (kernel-written, cleared by f3dgs_forward_counts)
    uint32_t* dev = nullptr;        // the same words as the current device sees them (written by a kernel)
    hipEvent_t done = nullptr;
    int device = -1;
    hipStream_t stream = nullptr;
    bool in_graph = false;          // a captured graph holds `dev`: the slot is never recycled
};
// What this thread last read of a device's counts: the provision of the next sync-free frame (option instance_capacity = 0) and the
// num_rendered a captured call returns (f3dgs.h: the count of a frame replayed from a graph is read with f3dgs_forward_counts).
struct CountHint { bool known = false; uint32_t own = 0, ref = 0; };
thread_local CountHint g_count_hint[64];
thread_local const uint32_t* g_last_forward_words = nullptr;
bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusActive;
}
// A stream that is being captured into a graph must not allocate: every eager call leaves one SPARE slot per device behind
// (stream = kSpareSlot), which the first captured call on a new stream takes over.
const hipStream_t kSpareSlot = reinterpret_cast<hipStream_t>(~uintptr_t(0));
bool make_count_slot(std::vector<CountReadback>& slots, int dev, hipStream_t s) {
    CountReadback rb;
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return false; }
    rb.host = static_cast<uint32_t*>(p);
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess || hipEventCreateWithFlags(&rb.done, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(p);
        return false;
    }
    rb.dev = static_cast<uint32_t*>(d);
    rb.device = dev;
    rb.stream = s;
    memset(p, 0, 64);
    if (slots.size() >= 64) {       // a caller that churns through streams: recycle the oldest slot no graph holds
        for (size_t i = 0; i < slots.size(); i++)
            if (!slots[i].in_graph && slots[i].stream != kSpareSlot) {
                (void)hipEventDestroy(slots[i].done);
                (void)hipHostFree(slots[i].host);
                slots.erase(slots.begin() + (long)i);
                break;
            }
    }
    slots.push_back(rb);
    return true;
}
// (the returned pointer is good until this thread's next call of this function)
CountReadback* count_readback(hipStream_t s, bool may_create) {
    thread_local std::vector<CountReadback> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    auto find = [&](hipStream_t key) -> CountReadback* {
        for (CountReadback& rb : slots)
            if (rb.device == dev && rb.stream == key) return &rb;
        return nullptr;
    };
    if (!may_create) {      // (a stream under capture: no allocation, no event creation)
        if (CountReadback* rb = find(s)) return rb;
        if (CountReadback* spare = find(kSpareSlot)) { spare->stream = s; return spare; }
        return nullptr;
    }
    if (!find(kSpareSlot)) (void)make_count_slot(slots, dev, kSpareSlot);      // (a missing spare only matters to a later capture)
    if (!find(s) && !make_count_slot(slots, dev, s)) return nullptr;
    return find(s);
}
}  // namespace

extern "C" {

int f3dgs_version(void) { return 30500; }   // 3.5.0 (major * 10000 + minor * 100 + patch): 3.5 options sync_free / instance_capacity, graph capture, f3dgs_forward_counts; 3.1 seven untested shape knobs removed, f3dgs_option_name; 3.2 f3dgs_set_feature_grad_lowres; 3.3 option bwd_bf16, 16-byte alignment checked; 3.4 bwd_bf16 = -1 (by the frame's conditioning), f3dgs_last_backward_contraction

int f3dgs_last_backward_contraction(void) { return g_last_bwd_bf16.load(); }

const uint32_t* f3dgs_forward_counts(void) { return g_last_forward_words; }

int f3dgs_set_option(const char* name, int value) {
    if (!name) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null option name");
    for (const OptionDesc& d : kOptions)
        if (strcmp(d.name, name) == 0) {
            options().*(d.field) = value;
            return F3DGS_OK;
        }
    return fail(F3DGS_ERR_INVALID_ARGUMENT, "unknown option '%s'", name);
}

int f3dgs_get_option(const char* name, int* value) {
    if (!name || !value) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null argument");
    for (const OptionDesc& d : kOptions)
        if (strcmp(d.name, name) == 0) {
            *value = options().*(d.field);
            return F3DGS_OK;
        }
    return fail(F3DGS_ERR_INVALID_ARGUMENT, "unknown option '%s'", name);
}

const char* f3dgs_option_name(int index) {
    const int n = (int)(sizeof(kOptions) / sizeof(kOptions[0]));
    return (index >= 0 && index < n) ? kOptions[index].name : nullptr;
}

const char* f3dgs_last_error(void) { return g_err.c_str(); }

void f3dgs_set_feature_grad_ready_callback(f3dgs_stage_fn fn, void* ctx) {
    g_feature_ready_fn = fn;
    g_feature_ready_ctx = ctx;
}

void f3dgs_set_feature_grad_accumulate(int on) { g_feature_accumulate = on ? 1 : 0; }

int f3dgs_set_feature_grad_lowres(const float* gx, int Hg, int Wg, const float* scale) {
    if (!gx) { g_lowres = LowresGrad(); return F3DGS_OK; }
    if (Hg <= 0 || Wg <= 0) return fail(F3DGS_ERR_INVALID_ARGUMENT, "bad low-resolution size %d x %d", Hg, Wg);
    g_lowres.gx = gx; g_lowres.scale = scale; g_lowres.Hg = Hg; g_lowres.Wg = Wg;
    return F3DGS_OK;
}

int f3dgs_debug_band_order(int gx, int gy, int tile_row_begin, int tile_row_end, uint32_t* tiles_out) {
    if (gx <= 0 || gy <= 0 || !tiles_out) return fail(F3DGS_ERR_INVALID_ARGUMENT, "bad grid");
    const int b0r = std::min(std::max(tile_row_begin, 0), gy), b1r = std::min(std::max(tile_row_end, b0r), gy);
    uint32_t b0 = 0, tb = 0;
    band_perm_params(gx, gy, b0r, b1r, &b0, &tb);
    const uint32_t T = (uint32_t)gx * (uint32_t)gy;
    for (uint32_t v = 0; v < T; v++) tiles_out[v] = band_perm(v, T, b0, tb);
    return tb != 0 ? 1 : 0;
}

void f3dgs_set_tile_band(int tile_row_begin, int tile_row_end) {
    g_band_begin = tile_row_begin;
    g_band_end = tile_row_end;
}

void f3dgs_set_grad_rows_ready_callback(f3dgs_rows_fn fn, void* ctx, int chunks) {
    g_rows_ready_fn = fn;
    g_rows_ready_ctx = ctx;
    g_rows_ready_chunks = chunks > 0 ? chunks : 1;
}

int f3dgs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                       uint8_t* present, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (P < 0) return fail(F3DGS_ERR_INVALID_ARGUMENT, "P < 0");
    if (P == 0) return F3DGS_OK;
    if (!means3D || !viewmatrix || !present) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null pointer");
    (void)projmatrix;
    launch_mark_visible(P, means3D, viewmatrix, present, s);
    HIP_TRY(hipGetLastError());
    return F3DGS_OK;
}

int f3dgs_forward(f3dgs_resize_fn geometry_resize, void* geometry_ctx, f3dgs_resize_fn binning_resize,
                  void* binning_ctx, f3dgs_resize_fn image_resize, void* image_ctx, int P, int D, int M, int C,
                  const float* background, int width, int height, const float* means3D, const float* shs,
                  const float* colors_precomp, const float* semantic_feature, const float* opacities,
                  const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                  const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                  float tan_fovy, int prefiltered, float* out_color, float* out_feature_map, float* out_depth,
                  int* radii, int debug, void* stream, int* num_rendered) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)prefiltered;  // the reference only uses it to __trap() on an impossible cull (auxiliary.h:162-166)
    if (num_rendered) *num_rendered = 0;
    if (P < 0 || C < 0 || width <= 0 || height <= 0) return fail(F3DGS_ERR_INVALID_ARGUMENT, "bad sizes");
    if (!geometry_resize || !binning_resize || !image_resize) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null resize hook");
    if (!out_color || !out_depth || (C > 0 && !out_feature_map)) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null output");
    const size_t HW = (size_t)width * height;
    if (P == 0) {  // rasterize_points.cu:84: outputs stay zero
        const bool cap0 = stream_is_capturing(s);      // (inside a graph: a kernel node, see zero_fill)
        HIP_TRY(zero_fill(out_color, 3 * HW * sizeof(float), cap0, s));
        HIP_TRY(zero_fill(out_depth, HW * sizeof(float), cap0, s));
        if (C) HIP_TRY(zero_fill(out_feature_map, (size_t)C * HW * sizeof(float), cap0, s));
        return F3DGS_OK;
    }
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !background)
        return fail(F3DGS_ERR_INVALID_ARGUMENT, "null input");
    if (!colors_precomp && !shs) return fail(F3DGS_ERR_INVALID_ARGUMENT, "need shs or colors_precomp");
    if (!colors_precomp && !cam_pos) return fail(F3DGS_ERR_INVALID_ARGUMENT, "SH colours need cam_pos");
    if (!cov3D_precomp && (!scales || !rotations))
        return fail(F3DGS_ERR_INVALID_ARGUMENT, "need scales+rotations or cov3D_precomp");
    if (C > 0 && !semantic_feature) return fail(F3DGS_ERR_INVALID_ARGUMENT, "semantic_feature is null but C > 0");
    if (shs && (D < 0 || D > 3 || (D + 1) * (D + 1) > M))
        return fail(F3DGS_ERR_INVALID_ARGUMENT, "SH degree %d needs %d coefficients, M = %d", D, (D + 1) * (D + 1), M);
    if (misaligned16({shs, rotations, semantic_feature}))
        return fail(F3DGS_ERR_INVALID_ARGUMENT, "shs, rotations and semantic_feature must be 16-byte aligned (f3dgs.h, Alignment)");

    int rc = F3DGS_OK;
    ViewParams vp;
    fill_view(vp, viewmatrix, projmatrix, cam_pos, background, tan_fovx, tan_fovy, width, height, scale_modifier);
    // (one part in a thousand of slack on the ratio: a scene built with scales of exactly 16 : 1 stays on the bf16 side)
    vp.max_axis_ratio = 1.001f * (float)options().bwd_bf16_max_ratio;
    if (g_band_begin != 0 || g_band_end != 0) {      // a tile band of the view (f3dgs.h: f3dgs_set_tile_band); begin >= end: an empty one
        vp.band0 = std::min(std::max(g_band_begin, 0), vp.gy);
        vp.band1 = std::min(std::max(g_band_end, vp.band0), vp.gy);
    }
    const int2 band = make_int2(vp.band0, vp.band1);
    const size_t tiles = (size_t)vp.gx * vp.gy;

    size_t geom_bytes = 0, img_bytes = 0, bin_bytes = 0;
    GeomState::carve(nullptr, P, &geom_bytes);
    char* geom_ptr = geometry_resize(geometry_ctx, geom_bytes);
    if (!geom_ptr) return fail(F3DGS_ERR_ALLOC, "geometry buffer allocation of %zu bytes failed", geom_bytes);
    GeomState geom = GeomState::carve(geom_ptr, P, nullptr);
    ImageState::carve(nullptr, HW, tiles, &img_bytes);
    char* img_ptr = image_resize(image_ctx, img_bytes);
    if (!img_ptr) return fail(F3DGS_ERR_ALLOC, "image buffer allocation of %zu bytes failed", img_bytes);
    ImageState img = ImageState::carve(img_ptr, HW, tiles, nullptr);
    // the single-pass sorts handle tile ids of up to two 8-bit digits (65536 tiles = 4096 x 4096 pixels and beyond
    // 4K); larger grids take the three-kernel passes
    const bool onesweep = options().sort_onesweep != 0 && tiles <= 65536;

    // Option sync_free / a stream under graph capture (f3dgs.h, "Sync-free forward"): the host does not wait for the instance
    // count between the depth sort and the emission.  The binning buffer is carved for a CAPACITY, the emit kernel and the tile
    // sort take the count from the device's word, and the count is read behind the LAST launch of the call (too small a
    // provision: the binning and the blend run once more with the exact length).  Under capture nothing is read at all.
    const bool capturing = stream_is_capturing(s);
    // sync_free: 1 always, 0 never (a capture is refused), -1 (default): inside a capture only.  Eager, the wait it removes was
    // already hidden behind the depth sort: an A / B in one process (tools/sync_free_ab.py) reads 0.1925 / 0.1912 ms per step at c1 and
    // 1.1315 / 1.1337 at c2 with the option off / on - nothing; what the option buys is the capture (c1: 0.19 -> 0.12 ms).
    const int sf_opt = options().sync_free;
    const bool sync_free = !onesweep && (sf_opt > 0 || (sf_opt < 0 && capturing));
    if (capturing && !sync_free)
        return fail(F3DGS_ERR_UNSUPPORTED, "the stream is being captured into a graph: the forward call reads the instance count on the host "
                    "unless option sync_free is 1 or -1 (and sort_onesweep = 0)");
    if (capturing && debug) return fail(F3DGS_ERR_UNSUPPORTED, "debug = 1 synchronises after every stage: not inside a graph capture");
    int dev_index = 0;
    HIP_TRY(hipGetDevice(&dev_index));
    if (dev_index < 0 || dev_index >= 64) return fail(F3DGS_ERR_UNSUPPORTED, "device index %d", dev_index);
    CountHint& hint = g_count_hint[dev_index];

    StageTimer tm(s, !capturing);
    // K1: projection, culling, SH colour, tile counts
    const int cull = tile_cull_enabled();
    launch_preprocess(P, D, M, means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp, vp, radii,
                      geom, cull, s);
    if ((rc = check_debug(debug, s, "preprocess"))) return rc;
    tm.mark("preprocess");
    if (onesweep) launch_sort_prologue(geom, (size_t)P, s);   // digit histograms of the depth keys + both instance totals
    // Both instance totals are final here.  Their read-back (the counterpart of rasterizer_impl.cu:283) is
    // requested now and awaited only after the depth sort has been enqueued, so the host round
    // trip hides behind ~0.1 ms of GPU work instead of idling the device.
    CountReadback* rbp = count_readback(s, !capturing);
    if (!rbp)
        return capturing ? fail(F3DGS_ERR_UNSUPPORTED, "graph capture: run one forward call on this stream (same host thread) before capturing - "
                                "the pinned count words are allocated on a stream's first call")
                         : fail(F3DGS_ERR_ALLOC, "pinned read-back buffer / event creation failed");
    CountReadback& rb = *rbp;
    if (capturing) rb.in_graph = true;
    g_last_forward_words = rb.host;
    if (onesweep) {
        HIP_TRY(hipMemcpyAsync(rb.host, geom.counters, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(rb.done, s));
    }

    // depth sort of the Gaussians (ids start in index order -> ties keep ascending id).  Three-kernel flavour: its
    // first kernel adds up the totals and stores them into the pinned host words itself (no totals / copy launches);
    // rb.done is recorded right behind that kernel.
    const TotalsJob tj = {geom.ref_partial, (P + 255) / 256, geom.counters, rb.dev, capturing ? nullptr : rb.done};
    const OffsetSumsJob sums = {geom.tiles_touched, geom.scan_tmp, geom.scan_sub};
    bool sums_done = false;         // small scenes: the one-workgroup depth sort leaves the list-offset sums as well
    if (onesweep) launch_depth_sort_onesweep(geom, (size_t)P, s);
    else HIP_TRY(launch_depth_sort(geom.depth_key, geom.key_a, geom.val_a, geom.key_b, geom.val_b, (size_t)P, geom.hist, &tj, s, &sums, &sums_done));
    if ((rc = check_debug(debug, s, "depth sort"))) return rc;
    tm.mark("depth_sort");
    const uint32_t* order = geom.val_a;

    // instance offsets in depth order (three-kernel flavour only; the single-pass emit scans on the fly)
    if (!onesweep && !sums_done)
        launch_offset_sums(geom.tiles_touched, order, (size_t)P, geom.scan_tmp, geom.scan_sub, s);

    // [0] instances in our lists, [1] the reference's bounding-rectangle count,
    // [2] != 0: a visible Gaussian of the frame is longer than bwd_bf16_max_ratio times its width (the single-pass flavour of the
    // binning does not produce the word: "unknown" reads as "yes")
    bool known = false;           // the host holds this frame's counts
    uint32_t N = 0, n_ref = 0, cap = 0;
    auto read_counts = [&]() -> int {
        HIP_TRY(hipEventSynchronize(rb.done));
        N = rb.host[0]; n_ref = rb.host[1];
        note_frame(geom_ptr, (onesweep || rb.host[2] != 0u) ? 1.f : 0.f, vp.band0, vp.band1);
        hint.known = true; hint.own = N; hint.ref = n_ref;
        known = true;
        if (N >= (1u << 30) || n_ref >= (1u << 31)) return fail(F3DGS_ERR_UNSUPPORTED, "more than 2^30 instances");
        return F3DGS_OK;
    };
    if (sync_free && (options().instance_capacity > 0 || hint.known)) {
        unsigned long long want = options().instance_capacity > 0 ? (unsigned long long)options().instance_capacity
                                                                  : (unsigned long long)hint.own + hint.own / 4 + 4096ull;
        // a list that fits the one-launch tile sort keeps a provision that does (the launches are chosen by the provision: a
        // replayed c1 step took the three-launch pass, 23 us, for 12,619 entries because it provided for 19,869)
        if (options().instance_capacity <= 0 && hint.own <= (uint32_t)SMALL_SORT_MAX && want > (unsigned long long)SMALL_SORT_MAX)
            want = SMALL_SORT_MAX;
        cap = (uint32_t)std::min(want, (1ull << 30) - 1ull);
    } else if (capturing) {
        return fail(F3DGS_ERR_UNSUPPORTED, "graph capture: no provision for the instance lists - run one forward call on this thread and device "
                    "before capturing, or set option instance_capacity");
    } else {      // the blocking read of the reference (and a sync-free thread's first frame on a device)
        if ((rc = read_counts())) return rc;
        cap = N;
    }
    tm.mark("scan+sync");

    const int bits = tile_bits((int)tiles);
    const int passes = (bits + RADIX_BITS - 1) / RADIX_BITS;
    BinState bin;
    for (;;) {
        const uint32_t n_carve = known ? N : cap;           // entries the binning buffer holds in this round
        BinState::carve(nullptr, n_carve, &bin_bytes, tiles);
        char* bin_ptr = binning_resize(binning_ctx, bin_bytes);
        if (!bin_ptr) return fail(F3DGS_ERR_ALLOC, "binning buffer allocation of %zu bytes failed", bin_bytes);
        bin = BinState::carve(bin_ptr, n_carve, nullptr, tiles);
        rb.host[3] = n_carve;
        // result must land in (tile_sorted, point_list) == the "A" side
        uint32_t* in_tile = (passes % 2 == 0) ? bin.tile_sorted : bin.tile_tmp;
        uint32_t* in_id = (passes % 2 == 0) ? bin.point_list : bin.id_tmp;
        if (onesweep) {
            HIP_TRY(hipMemsetAsync(img.tile_len, 0, tiles * sizeof(uint32_t), s));
            // emits, builds the tile digit histograms, presets the ranges
            launch_emit_scan(P, geom, bin, order, vp.gx, vp.gy, band, cull, in_tile, in_id, N, bin.ranges_enc, s);
            if ((rc = check_debug(debug, s, "emit"))) return rc;
            tm.mark("emit");
            launch_tile_sort_onesweep(geom, bin, N, passes, bin.ranges_enc, s);   // the final pass also records the tile ranges
            if ((rc = check_debug(debug, s, "tile sort"))) return rc;
            tm.mark("tile_sort");
        } else if (known && N == 0) {
            // all-ones = "no entry yet" for both halves of the encoded ranges (see BinState::ranges_enc)
            HIP_TRY(hipMemsetAsync(bin.ranges_enc, 0xFF, tiles * sizeof(uint2), s));
            HIP_TRY(hipMemsetAsync(img.tile_len, 0, tiles * sizeof(uint32_t), s));
        } else {
            // presets the ranges too; a captured frame that finds no room raises the slot's sticky word (the host of an eager frame
            // learns it from the count)
            launch_emit_instances(P, geom, order, vp.gx, vp.gy, band, cull, in_tile, in_id, bin.ranges_enc, img.tile_len, n_carve,
                                  known ? nullptr : geom.counters, capturing ? rb.dev + 4 : nullptr, s);
            if ((rc = check_debug(debug, s, "emit"))) return rc;
            tm.mark("emit");
            // the final pass also records the tile ranges
            launch_radix_sort_pairs(bin.tile_sorted, bin.point_list, bin.tile_tmp, bin.id_tmp, n_carve, bits, bin.hist, true,
                                    bin.ranges_enc, s, known ? nullptr : geom.counters);
            if ((rc = check_debug(debug, s, "tile sort"))) return rc;
            tm.mark("tile_sort");
        }

        launch_render_forward(vp, C, bin.ranges_enc, img.ranges, bin.point_list, geom.rec, semantic_feature, img.final_T,
                              img.n_contrib, out_color, out_feature_map, out_depth, img.tile_len, s);
        if ((rc = check_debug(debug, s, "render"))) return rc;
        tm.mark("render_fwd");
        if (known || capturing) break;
        // sync-free: the count is read HERE, behind the last launch of the call (it was final right behind the first kernel of
        // the depth sort, so the wait is over by now); a frame that found no room is run again, with the exact length
        if ((rc = read_counts())) return rc;
        if (N <= cap) break;
    }
    if (capturing) {
        // nothing of this frame is known to the host: the blend backward of bwd_bf16 = -1 lets the DEVICE's long-axis word pick its
        // first window (note 2: both shapes are launched, one runs), and the caller gets the last count this thread read on the
        // device (>= 1: the backward call only asks whether anything was listed)
        note_frame(geom_ptr, 2.f, vp.band0, vp.band1);
        n_ref = std::max(1u, hint.ref);
    }
    if (num_rendered) *num_rendered = (int)n_ref;
    HIP_TRY(hipGetLastError());
    return F3DGS_OK;
}

size_t f3dgs_backward_scratch_bytes(int P, int C) {
    (void)C;
    return ((size_t)(P > 0 ? P : 0) * GREC * sizeof(float) + ALIGN - 1) & ~(ALIGN - 1);
}

int f3dgs_backward(int P, int D, int M, int C, int R, const float* background, int width, int height,
                   const float* means3D, const float* shs, const float* colors_precomp,
                   const float* semantic_feature, const float* scales, float scale_modifier, const float* rotations,
                   const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                   float tan_fovx, float tan_fovy, const int* radii, const char* geom_buffer,
                   const char* binning_buffer, const char* image_buffer, const float* dL_dpix,
                   const float* dL_dfeaturepix, const float* dL_depths, float* dL_dmean2D, float* dL_dconic,
                   float* dL_dopacity, float* dL_dcolor, float* dL_dsemantic_feature, float* dL_dmean3D,
                   float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dz, void* scratch,
                   int debug, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)semantic_feature;  // Q3: never read by the reference's backward either
    (void)colors_precomp;    // colours were copied into the splat records by the forward pass
    // f3dgs.h: a low-resolution gradient set by f3dgs_set_feature_grad_lowres is consumed by THIS call whatever it returns -
    // taken before the first return, so that a pointer the caller may free after a failed or empty call never stays armed
    const LowresGrad lowres = g_lowres;
    g_lowres = LowresGrad();
    if (P < 0 || C < 0 || R < 0) return fail(F3DGS_ERR_INVALID_ARGUMENT, "bad sizes");
    if (P == 0) return F3DGS_OK;
    if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer))
        return fail(F3DGS_ERR_INVALID_ARGUMENT, "null state buffer");
    if (!radii) return fail(F3DGS_ERR_INVALID_ARGUMENT, "radii is required");
    if (lowres.gx) {
        if (C == 0) return fail(F3DGS_ERR_INVALID_ARGUMENT, "a low-resolution feature-map gradient was set but C = 0");
        if (lowres.Hg > height || lowres.Wg > width)
            return fail(F3DGS_ERR_UNSUPPORTED, "low-resolution feature-map gradient %d x %d is larger than the image %d x %d: "
                        "apply the transposed resize outside (f3dgs_feature_l1 with d_feature_map)", lowres.Hg, lowres.Wg, height, width);
        if (!options().feature_mfma)
            return fail(F3DGS_ERR_UNSUPPORTED, "the low-resolution feature-map gradient is taken by the pixel-lane backward (option feature_mfma = 1)");
    }
    if (!dL_dpix || !dL_depths || (C > 0 && !dL_dfeaturepix && !lowres.gx)) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null upstream grad");
    if (!dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dcov3D || (C > 0 && !dL_dsemantic_feature))
        return fail(F3DGS_ERR_INVALID_ARGUMENT, "null output");
    if (M > 0 && shs && !dL_dsh) return fail(F3DGS_ERR_INVALID_ARGUMENT, "dL_dsh is null");
    if (scales && (!dL_dscale || !dL_drot || !rotations)) return fail(F3DGS_ERR_INVALID_ARGUMENT, "scale/rot grads null");
    if (!scratch) return fail(F3DGS_ERR_INVALID_ARGUMENT, "scratch is null");
    if (misaligned16({shs, rotations, dL_dsh, dL_drot, dL_dconic, dL_dsemantic_feature, scratch}))
        return fail(F3DGS_ERR_INVALID_ARGUMENT, "shs, rotations, dL_dsh, dL_drot, dL_dconic, dL_dsemantic_feature and scratch must be "
                    "16-byte aligned (f3dgs.h, Alignment)");

    int rc = F3DGS_OK;
    ViewParams vp;
    fill_view(vp, viewmatrix, projmatrix, campos, background, tan_fovx, tan_fovy, width, height, scale_modifier);
    const size_t HW = (size_t)width * height, tiles = (size_t)vp.gx * vp.gy;
    GeomState geom = GeomState::carve(const_cast<char*>(geom_buffer), P, nullptr);
    // R is the reference-style count, NOT the length the forward carved the binning buffer with; the backward
    // only reads the sorted instance list, which BinState keeps at offset 0 whatever the length.
    const uint32_t* point_list = BinState::list_of(binning_buffer);
    ImageState img = ImageState::carve(const_cast<char*>(image_buffer), HW, tiles, nullptr);
    float* grec = static_cast<float*>(scratch);

    const bool capturing = stream_is_capturing(s);      // (a step replayed from a graph: no events, no debug synchronisation)
    if (capturing && debug) return fail(F3DGS_ERR_UNSUPPORTED, "debug = 1 synchronises after every stage: not inside a graph capture");
    StageTimer tm(s, !capturing);
    HIP_TRY(zero_fill(grec, (size_t)P * GREC * sizeof(float), capturing, s));
    // (f3dgs_set_feature_grad_accumulate: the caller's buffer already holds the sum over its earlier views)
    // Clearing this buffer ahead of time on a side stream - under the blend forward, forked right in front of it - was measured
    // at c4 / c5 in round 5: the fill leaves the backward pass (0.32 -> 0.02 ms) and costs the forward blend MORE (1.95 -> 2.41 ms
    // at c4, 3.40 -> 4.07 at c5: its feature-row gathers queue behind 2 - 2.5 GB of writes); profiles/r05_notes.md.
    if (C > 0 && !g_feature_accumulate) HIP_TRY(zero_fill(dL_dsemantic_feature, (size_t)P * C * sizeof(float), capturing, s));
    tm.mark("zero");
    // Contraction precision of the blend backward (option bwd_bf16: 1 two-term bf16, 0 exact fp32, -1 by the frame): the
    // covariance chain behind the blend (backward.cu:144-341) amplifies an error of the blend-level sums by the square of a
    // Gaussian's axis ratio; measured (profiles/r06_ratio_sweep.txt) the bf16 shape stays within a third of the gradient bound of
    // the exact shape up to a ratio of 16 and leaves it from 32 on - where two runs of the EXACT shape differ by more than a
    // bound as well.  So: bf16 while no visible Gaussian of the frame is longer than bwd_bf16_max_ratio (default 16) times its
    // width, exact fp32 otherwise - and exact when the forward call's note is gone.
    // 1: bf16 two-term everywhere; 2: hybrid - bf16 feature / colour blocks, the moment block (what the chain amplifies) on exact
    // fp32 matrix instructions; 0: exact fp32 everywhere (option bwd_bf16 = 0 only)
    int contraction = options().bwd_bf16 > 0 ? 1 : 0;
    {
        float long_axis = 1.f;       // what the forward call of this frame noted (against bwd_bf16_max_ratio as it was THEN)
        int b0 = 0, b1 = vp.gy;      // ... and the tile rows it listed (a note that is gone: the whole grid - right, only unbalanced)
        const bool noted = frame_axis_ratio(geom_buffer, &long_axis, &b0, &b1);
        if (noted) { vp.band0 = std::min(std::max(b0, 0), vp.gy); vp.band1 = std::min(std::max(b1, vp.band0), vp.gy); }
        if (options().bwd_bf16 < 0) {
            contraction = (noted && long_axis == 0.f) ? 1 : 2;
            if (noted && long_axis == 2.f) contraction = 3;     // a captured frame: decided on the device (GeomState::counters[2])
        }
    }
    if (R > 0) {
        const int ran = launch_render_backward(vp, C, img.ranges, point_list, geom.rec, img.final_T, img.n_contrib,
                                               dL_dpix, dL_dfeaturepix, dL_depths, grec, dL_dsemantic_feature, img.tile_len, img.tile_order,
                                               lowres.gx ? &lowres : nullptr, contraction, geom.counters + 2, s);
        g_last_bwd_bf16.store(ran);
    }
    if ((rc = check_debug(debug, s, "render backward"))) return rc;
    tm.mark("render_bwd");
    if (g_feature_ready_fn) g_feature_ready_fn(g_feature_ready_ctx, stream);   // dL_dsemantic_feature is final on `s` here
    // The per-Gaussian stage, in `chunks` launches over consecutive row ranges when a rows-ready callback asks for it:
    // behind launch k every per-Gaussian gradient of rows [r0, r1) is final on `s` (the kernel is row-parallel), so a
    // data-parallel caller can start reducing them - the SH gradient is 3/4 of the non-feature message - while the
    // later chunks still run.
    {
        const int align = preprocess_backward_row_align();
        int chunks = g_rows_ready_fn ? std::max(1, g_rows_ready_chunks) : 1;
        int per = (int)(((long long)P + chunks - 1) / chunks);
        per = std::max(align, (per + align - 1) / align * align);
        for (int r0 = 0; r0 < P; r0 += per) {
            const int r1 = std::min(P, r0 + per);
            launch_preprocess_backward(P, D, M, C, means3D, radii, shs, scales, rotations, cov3D_precomp, vp, geom, grec,
                                       dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D,
                                       (M > 0 && shs) ? dL_dsh : nullptr, scales ? dL_dscale : nullptr,
                                       scales ? dL_drot : nullptr, dL_dz, r0, r1, s);
            if (g_rows_ready_fn) g_rows_ready_fn(g_rows_ready_ctx, stream, r0, r1);
        }
    }
    if ((rc = check_debug(debug, s, "preprocess backward"))) return rc;
    tm.mark("preprocess_bwd");
    HIP_TRY(hipGetLastError());
    return F3DGS_OK;
}

size_t f3dgs_feature_l1_scratch_bytes(int C, int Cout, int Hg, int Wg, int has_decoder) {
    if (C <= 0 || Cout <= 0 || Hg <= 0 || Wg <= 0) return 0;
    return feature_l1_scratch_bytes(C, Cout, Hg, Wg, has_decoder != 0);
}

int f3dgs_feature_l1(int C, int H, int W, int Cout, int Hg, int Wg, const float* feature_map, const float* weight,
                     const float* bias, const float* gt, float* loss, float* d_feature_map, float* d_weight,
                     float* d_bias, void* scratch, void* stream) {
    if (C <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Hg <= 0 || Wg <= 0) return fail(F3DGS_ERR_INVALID_ARGUMENT, "bad sizes");
    if (!feature_map || !gt || !loss || !scratch) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null pointer");
    if ((weight == nullptr) != (bias == nullptr)) return fail(F3DGS_ERR_INVALID_ARGUMENT, "weight and bias go together");
    if (weight) {
        if (!d_weight || !d_bias) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null decoder gradient");
        if (!feature_l1_decoder_supported(C))
            return fail(F3DGS_ERR_UNSUPPORTED, "decoder input width %d: supported are 32, 64, 128", C);
    } else if (Cout != C) {
        return fail(F3DGS_ERR_INVALID_ARGUMENT, "without a decoder the ground truth must have C = %d channels, got %d", C, Cout);
    }
    if ((long long)Hg * Wg * (long long)(Cout > C ? Cout : C) >= (1ll << 40)) return fail(F3DGS_ERR_UNSUPPORTED, "too large");
    HIP_TRY(launch_feature_l1(C, H, W, Cout, Hg, Wg, feature_map, weight, bias, gt, loss, d_feature_map, d_weight, d_bias,
                              static_cast<char*>(scratch), static_cast<hipStream_t>(stream)));
    return F3DGS_OK;
}

const float* f3dgs_feature_l1_lowres_grad(int C, int Cout, int Hg, int Wg, int has_decoder, void* scratch) {
    if (C <= 0 || Cout <= 0 || Hg <= 0 || Wg <= 0 || !scratch) return nullptr;
    return feature_l1_lowres_grad(static_cast<char*>(scratch), C, Cout, Hg, Wg, has_decoder != 0);
}

size_t f3dgs_feature_decode_scratch_bytes(int C, int Hg, int Wg, int has_decoder) {
    if (C <= 0 || Hg <= 0 || Wg <= 0) return 0;
    return feature_decode_scratch_bytes(C, Hg, Wg, has_decoder != 0);
}

int f3dgs_feature_decode(int C, int H, int W, int Cout, int Hg, int Wg, const float* feature_map, const float* weight,
                         const float* bias, void* out, int out_is_half, void* scratch, void* stream) {
    if (C <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Hg <= 0 || Wg <= 0) return fail(F3DGS_ERR_INVALID_ARGUMENT, "bad sizes");
    if (!feature_map || !out) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null pointer");
    if ((weight == nullptr) != (bias == nullptr)) return fail(F3DGS_ERR_INVALID_ARGUMENT, "weight and bias go together");
    if (weight) {
        if (!scratch) return fail(F3DGS_ERR_INVALID_ARGUMENT, "scratch is null");
        if (!feature_l1_decoder_supported(C))
            return fail(F3DGS_ERR_UNSUPPORTED, "decoder input width %d: supported are 32, 64, 128", C);
    } else if (Cout != C) {
        return fail(F3DGS_ERR_INVALID_ARGUMENT, "without a decoder the output has C = %d channels, got %d", C, Cout);
    }
    HIP_TRY(launch_feature_decode(C, H, W, Cout, Hg, Wg, feature_map, weight, bias, out, out_is_half != 0,
                                  static_cast<char*>(scratch), static_cast<hipStream_t>(stream)));
    return F3DGS_OK;
}

int f3dgs_adam_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                    double beta2, double eps, int step, void* stream) {
    return f3dgs_adam_step_rows(n, 1, nullptr, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, stream);
}

int f3dgs_adam_step_rows(size_t n, size_t width, const uint8_t* row_mask, float* param, const float* grad, float* exp_avg,
                         float* exp_avg_sq, double lr, double beta1, double beta2, double eps, int step, void* stream) {
    if (n == 0) return F3DGS_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null pointer");
    if (step < 1) return fail(F3DGS_ERR_INVALID_ARGUMENT, "step counts from 1");
    if (row_mask && (width == 0 || n % width != 0 || width > 0xFFFFFFFFull))
        return fail(F3DGS_ERR_INVALID_ARGUMENT, "n = %zu is not a whole number of rows of %zu floats", n, width);
    launch_adam_step(n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, row_mask, width, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return F3DGS_OK;
}

int f3dgs_adam_step_multi(int n_tensors, const f3dgs_adam_tensor* tensors, double beta1, double beta2, double eps,
                          const uint8_t* row_mask, size_t rows, void* stream) {
    if (n_tensors < 0 || n_tensors > F3DGS_ADAM_MAX_TENSORS) return fail(F3DGS_ERR_INVALID_ARGUMENT, "0 .. %d tensors per call", F3DGS_ADAM_MAX_TENSORS);
    if (n_tensors == 0) return F3DGS_OK;
    if (!tensors) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null table");
    size_t blocks = 0;
    for (int i = 0; i < n_tensors; i++) {
        const f3dgs_adam_tensor& t = tensors[i];
        if (t.n == 0) continue;
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) return fail(F3DGS_ERR_INVALID_ARGUMENT, "tensor %d: null pointer", i);
        if (t.step < 1) return fail(F3DGS_ERR_INVALID_ARGUMENT, "tensor %d: step counts from 1", i);
        blocks += ((t.n + 3) / 4 + 255) / 256;
    }
    if (blocks >= (1ull << 31)) return fail(F3DGS_ERR_UNSUPPORTED, "too large for one launch");
    if (row_mask && rows == 0) return fail(F3DGS_ERR_INVALID_ARGUMENT, "row mask without a row count");
    launch_adam_step_multi(n_tensors, tensors, beta1, beta2, eps, row_mask, rows, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return F3DGS_OK;
}

int f3dgs_densify_gather(size_t n_out, const int32_t* src_row, const uint8_t* kind, const int32_t* override_row, int n_tensors,
                         const f3dgs_densify_tensor* tensors, void* stream) {
    if (n_tensors < 0 || n_tensors > F3DGS_DENSIFY_MAX_TENSORS)
        return fail(F3DGS_ERR_INVALID_ARGUMENT, "n_tensors = %d: at most %d per call", n_tensors, F3DGS_DENSIFY_MAX_TENSORS);
    if (n_out == 0 || n_tensors == 0) return F3DGS_OK;
    if (!src_row || !kind || !tensors) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null pointer");
    for (int i = 0; i < n_tensors; i++) {
        const f3dgs_densify_tensor& t = tensors[i];
        if (!t.src || !t.dst || t.width < 1) return fail(F3DGS_ERR_INVALID_ARGUMENT, "tensor %d: null pointer or width < 1", i);
        if (t.width > 65536) return fail(F3DGS_ERR_UNSUPPORTED, "tensor %d: rows of %d floats (at most 65536)", i, t.width);
        if (t.mode < F3DGS_DENSIFY_COPY || t.mode > F3DGS_DENSIFY_OVERRIDE_CHILD)
            return fail(F3DGS_ERR_INVALID_ARGUMENT, "tensor %d: unknown mode %d", i, t.mode);
        if (t.mode == F3DGS_DENSIFY_OVERRIDE_CHILD && (!t.override_src || !override_row))
            return fail(F3DGS_ERR_INVALID_ARGUMENT, "tensor %d: OVERRIDE_CHILD needs override_src and override_row", i);
    }
    launch_densify_gather(n_out, src_row, kind, override_row, n_tensors, tensors, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return F3DGS_OK;
}

size_t f3dgs_knn_scratch_bytes(int P) { return knn_scratch_bytes((size_t)(P > 0 ? P : 0)); }

int f3dgs_knn_mean_dist2(int P, const float* points, float* mean_dist2, void* scratch, void* stream) {
    if (P < 0) return fail(F3DGS_ERR_INVALID_ARGUMENT, "P < 0");
    if (P == 0) return F3DGS_OK;
    if (!points || !mean_dist2 || !scratch) return fail(F3DGS_ERR_INVALID_ARGUMENT, "null pointer");
    launch_knn_mean_dist2(P, points, mean_dist2, static_cast<char*>(scratch), static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return F3DGS_OK;
}

int f3dgs_debug_read(const char* what, int P, int C, int R, int width, int height, const char* geom_buffer,
                     const char* binning_buffer, const char* image_buffer, void* host_dst, size_t dst_bytes,
                     void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)C;
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;
    const size_t HW = (size_t)width * height, tiles = (size_t)gx * gy;
    GeomState geom = GeomState::carve(const_cast<char*>(geom_buffer), P, nullptr);
    // the binning buffer was carved with the length of OUR instance list (counters[0]), not with R
    // ... or, by a sync-free forward call, for a capacity: counters[3] holds what the emit kernel was given
    uint32_t n_list = (uint32_t)R, n_carved = (uint32_t)R;
    if (P > 0 && geom_buffer) {
        uint32_t words[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(words, geom.counters, sizeof words, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        n_list = words[0];
        n_carved = words[3] ? words[3] : words[0];
        if (n_list > n_carved) n_list = n_carved;       // (a frame that found no room: nothing behind the capacity was written)
    }
    BinState bin = BinState::carve(const_cast<char*>(binning_buffer), n_carved, nullptr);
    ImageState img = ImageState::carve(const_cast<char*>(image_buffer), HW, tiles, nullptr);
    const std::string w(what);
    const void* src = nullptr;
    size_t bytes = 0;
    if (w == "rec") { src = geom.rec; bytes = (size_t)P * sizeof(SplatRec); }
    else if (w == "clamped") { src = geom.clamped; bytes = P; }
    else if (w == "tiles_touched") { src = geom.tiles_touched; bytes = (size_t)P * 4; }
    else if (w == "depth_key") { src = geom.depth_key; bytes = (size_t)P * 4; }
    else if (w == "order") { src = geom.val_a; bytes = (size_t)P * 4; }
    else if (w == "counters") { src = geom.counters; bytes = 16 * 4; }
    else if (w == "point_list") { src = bin.point_list; bytes = (size_t)n_list * 4; }
    else if (w == "tile_sorted") { src = bin.tile_sorted; bytes = (size_t)n_list * 4; }
    else if (w == "ranges") { src = img.ranges; bytes = tiles * 8; }
    else if (w == "final_T") { src = img.final_T; bytes = HW * 4; }
    else if (w == "n_contrib") { src = img.n_contrib; bytes = HW * 4; }
    else return fail(F3DGS_ERR_INVALID_ARGUMENT, "unknown debug item '%s'", what);
    if (bytes > dst_bytes) bytes = dst_bytes;
    if (bytes) {
        HIP_TRY(hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return F3DGS_OK;
}

int f3dgs_profile_read(const char** names, double* total_ms, long* calls, int max_stages) {
    resolve_pending();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto& t : g_totals) {
        if (n >= max_stages) break;
        names[n] = t.first;
        total_ms[n] = t.second.first;
        calls[n] = t.second.second;
        n++;
    }
    return n;
}

void f3dgs_profile_reset(void) {
    resolve_pending();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_totals.clear();
}

}  // extern "C"
