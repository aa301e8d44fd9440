// resize_taps.h - the bilinear resize of the feature-map loss (F.interpolate(mode='bilinear', align_corners=True),
// R/train.py:99-101) as per-axis tap lists, shared by the loss kernels (feature_loss.hip) and by the blend backward when it
// takes the loss's gradient at the LOSS's resolution (render_bwd_pl.hip, f3dgs_set_feature_grad_lowres).
#pragma once

#include <hip/hip_runtime.h>

namespace f3dgs {

struct ResizeGeom {
    int H, W, Hg, Wg;
    float sy, sx;      // (in - 1) / (out - 1), 0 when out == 1 (PyTorch: area_pixel_compute_scale, align_corners)
};

inline ResizeGeom make_resize_geom(int H, int W, int Hg, int Wg) {
    ResizeGeom g;
    g.H = H; g.W = W; g.Hg = Hg; g.Wg = Wg;
    g.sy = Hg > 1 ? (float)(H - 1) / (float)(Hg - 1) : 0.f;
    g.sx = Wg > 1 ? (float)(W - 1) / (float)(Wg - 1) : 0.f;
    return g;
}

// source taps of output index o (PyTorch upsample_bilinear2d, align_corners = true)
__device__ __forceinline__ void taps(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    const float src = scale * (float)o;
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

// candidate outputs of a source index i: outputs o whose taps can include i
__device__ __forceinline__ void resize_cand(int i, float scale, int out, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = (i == 0) ? 0 : -1; return; }
    lo = max(0, (int)floorf((float)(i - 1) / scale) - 1);
    hi = min(out - 1, (int)ceilf((float)(i + 1) / scale) + 1);
}

// The transposed resize along one axis: the (output, weight) list of source index i, outputs ascending, zero weights
// dropped; returns the count, or -1 when it exceeds CAP.  Shrinking (in >= out) never needs more than two entries.
template <int CAP>
__device__ __forceinline__ int resize_build(int i, float scale, int in, int out, int* oo, float* ww) {
    int lo, hi, n = 0;
    resize_cand(i, scale, out, lo, hi);
    for (int o = lo; o <= hi; o++) {
        int a0, a1;
        float l0, l1;
        taps(o, scale, in, a0, a1, l0, l1);
        const float wv = (a0 == i ? l0 : 0.f) + (a1 == i ? l1 : 0.f);
        if (wv == 0.f) continue;
        if (n == CAP) return -1;
        oo[n] = o; ww[n] = wv; n++;
    }
    return n;
}

}  // namespace f3dgs
