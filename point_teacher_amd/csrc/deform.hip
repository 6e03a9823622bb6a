// (Modulated) deformable convolution for gfx950 - replaces mmcv.ops.DeformConv2d / ModulatedDeformConv2d
// (`conv_cfg=dict(type='DCNv2')`, reachable through `dcn_on_last_conv=True`:
// HBB_TOD/mmdet/models/dense_heads/anchor_free_head.py:101-102,121-122, fcos_head_p2b_ts.py:197-198).
//
// The contraction stays on the matrix cores through the vendor GEMM (north_star: "MFMA used only for the conv
// contractions"); what is hand-written is the data-dependent part around it:
// NCHW (first version, kept for NCHW callers):
//   deform_im2col        x, offset, mask      -> col[B, C*K, L]     (bilinear gather, one thread per (c, l): K taps)
//   deform_col2im        dcol, offset, mask   -> dx                 (scatter with f32 atomics)
//   deform_col2im_coord  dcol, x, offset, mask-> doffset, dmask     (one thread per offset element)
// channels_last (the training layout; second half of this file):
//   deform_im2col_cl     one wavefront per (pixel, tap): contiguous 1-KiB gathers and column stores
//   deform_col2im_cl     doffset / dmask by a wave reduction; dx of far samples by atomics
//   deform_col2im_gather dx of stride-1 layers without scatter (every input pixel collects its samples)
// Sampling rule of the published algorithm (mmcv modulated_deform_conv_cuda_kernel.cuh, dmcn_im2col_bilinear):
// a sample outside (-1, H) x (-1, W) is 0; inside, the four neighbours are weighted bilinearly and neighbours
// outside the map contribute 0.  Offsets are (dy, dx) interleaved per kernel tap, per deformable group.
#include "pt_common.h"

namespace pt {

struct DeformGeom {
  int B, C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, dg, Ho, Wo;
};

__device__ __forceinline__ float dcn_bilinear(const float* __restrict__ im, int H, int W, float h, float w) {
  if (!(h > -1.f && w > -1.f && h < (float)H && w < (float)W)) return 0.f;
  const int hl = (int)floorf(h), wl = (int)floorf(w), hh = hl + 1, wh = wl + 1;
  const float lh = h - (float)hl, lw = w - (float)wl, uh = 1.f - lh, uw = 1.f - lw;
  const float v1 = (hl >= 0 && wl >= 0) ? im[hl * W + wl] : 0.f;
  const float v2 = (hl >= 0 && wh <= W - 1) ? im[hl * W + wh] : 0.f;
  const float v3 = (hh <= H - 1 && wl >= 0) ? im[hh * W + wl] : 0.f;
  const float v4 = (hh <= H - 1 && wh <= W - 1) ? im[hh * W + wh] : 0.f;
  return uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4;
}

__global__ void __launch_bounds__(256)
    deform_im2col_kernel(const float* __restrict__ x, const float* __restrict__ offset, const float* __restrict__ mask,
                         DeformGeom g, float* __restrict__ col) {
  const int L = g.Ho * g.Wo, K = g.kh * g.kw;
  const long total = (long)g.B * g.C * L;
  const int cpg = g.C / g.dg;                                   // channels per deformable group
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int l = (int)(i % L), c = (int)((i / L) % g.C), b = (int)(i / ((long)L * g.C));
    const int ho = l / g.Wo, wo = l - ho * g.Wo, d = c / cpg;
    const float* im = x + ((size_t)b * g.C + c) * g.H * g.W;
    const float* off = offset + ((size_t)b * g.dg + d) * 2 * K * L;
    const float* mk = mask ? mask + ((size_t)b * g.dg + d) * K * L : nullptr;
    float* cp = col + ((size_t)b * g.C + c) * K * L + l;
    for (int t = 0; t < K; ++t) {
      const int ki = t / g.kw, kj = t - ki * g.kw;
      const float h = (float)(ho * g.sh - g.ph + ki * g.dh) + off[(size_t)(2 * t) * L + l];
      const float w = (float)(wo * g.sw - g.pw + kj * g.dw) + off[(size_t)(2 * t + 1) * L + l];
      float v = dcn_bilinear(im, g.H, g.W, h, w);
      if (mk) v *= mk[(size_t)t * L + l];
      cp[(size_t)t * L] = v;
    }
  }
}

__global__ void __launch_bounds__(256)
    deform_col2im_kernel(const float* __restrict__ dcol, const float* __restrict__ offset, const float* __restrict__ mask,
                         DeformGeom g, float* __restrict__ dx) {
  const int L = g.Ho * g.Wo, K = g.kh * g.kw;
  const long total = (long)g.B * g.C * K * L;
  const int cpg = g.C / g.dg;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int l = (int)(i % L), t = (int)((i / L) % K), c = (int)((i / ((long)L * K)) % g.C);
    const int b = (int)(i / ((long)L * K * g.C));
    const int ho = l / g.Wo, wo = l - ho * g.Wo, d = c / cpg, ki = t / g.kw, kj = t - ki * g.kw;
    const float* off = offset + ((size_t)b * g.dg + d) * 2 * K * L;
    const float h = (float)(ho * g.sh - g.ph + ki * g.dh) + off[(size_t)(2 * t) * L + l];
    const float w = (float)(wo * g.sw - g.pw + kj * g.dw) + off[(size_t)(2 * t + 1) * L + l];
    if (!(h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W)) continue;
    float gv = dcol[i];
    if (mask) gv *= mask[(((size_t)b * g.dg + d) * K + t) * L + l];
    if (gv == 0.f) continue;
    const int hl = (int)floorf(h), wl = (int)floorf(w), hh = hl + 1, wh = wl + 1;
    const float lh = h - (float)hl, lw = w - (float)wl, uh = 1.f - lh, uw = 1.f - lw;
    float* im = dx + ((size_t)b * g.C + c) * g.H * g.W;
    if (hl >= 0 && wl >= 0) atomicAdd(&im[hl * g.W + wl], gv * uh * uw);
    if (hl >= 0 && wh <= g.W - 1) atomicAdd(&im[hl * g.W + wh], gv * uh * lw);
    if (hh <= g.H - 1 && wl >= 0) atomicAdd(&im[hh * g.W + wl], gv * lh * uw);
    if (hh <= g.H - 1 && wh <= g.W - 1) atomicAdd(&im[hh * g.W + wh], gv * lh * lw);
  }
}

// One thread per (b, deformable group, tap, l): sums over the channels of the group.
__global__ void __launch_bounds__(256)
    deform_col2im_coord_kernel(const float* __restrict__ dcol, const float* __restrict__ x,
                               const float* __restrict__ offset, const float* __restrict__ mask, DeformGeom g,
                               float* __restrict__ doffset, float* __restrict__ dmask) {
  const int L = g.Ho * g.Wo, K = g.kh * g.kw;
  const long total = (long)g.B * g.dg * K * L;
  const int cpg = g.C / g.dg;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int l = (int)(i % L), t = (int)((i / L) % K), d = (int)((i / ((long)L * K)) % g.dg);
    const int b = (int)(i / ((long)L * K * g.dg));
    const int ho = l / g.Wo, wo = l - ho * g.Wo, ki = t / g.kw, kj = t - ki * g.kw;
    const float* off = offset + ((size_t)b * g.dg + d) * 2 * K * L;
    const float h = (float)(ho * g.sh - g.ph + ki * g.dh) + off[(size_t)(2 * t) * L + l];
    const float w = (float)(wo * g.sw - g.pw + kj * g.dw) + off[(size_t)(2 * t + 1) * L + l];
    const float m = mask ? mask[(((size_t)b * g.dg + d) * K + t) * L + l] : 1.f;
    float gh = 0.f, gw = 0.f, gm = 0.f;
    const bool inside = h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
    if (inside) {
      const int hl = (int)floorf(h), wl = (int)floorf(w), hh = hl + 1, wh = wl + 1;
      const float lh = h - (float)hl, lw = w - (float)wl, uh = 1.f - lh, uw = 1.f - lw;
      const bool a1 = hl >= 0 && wl >= 0, a2 = hl >= 0 && wh <= g.W - 1, a3 = hh <= g.H - 1 && wl >= 0,
                 a4 = hh <= g.H - 1 && wh <= g.W - 1;
      for (int cc = 0; cc < cpg; ++cc) {
        const int c = d * cpg + cc;
        const float* im = x + ((size_t)b * g.C + c) * g.H * g.W;
        const float v1 = a1 ? im[hl * g.W + wl] : 0.f, v2 = a2 ? im[hl * g.W + wh] : 0.f;
        const float v3 = a3 ? im[hh * g.W + wl] : 0.f, v4 = a4 ? im[hh * g.W + wh] : 0.f;
        const float gv = dcol[(((size_t)b * g.C + c) * K + t) * L + l];
        gh += gv * m * (uw * (v3 - v1) + lw * (v4 - v2));
        gw += gv * m * (uh * (v2 - v1) + lh * (v4 - v3));
        gm += gv * (uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4);
      }
    }
    doffset[(((size_t)b * g.dg + d) * 2 * K + 2 * t) * L + l] = gh;
    doffset[(((size_t)b * g.dg + d) * 2 * K + 2 * t + 1) * L + l] = gw;
    if (dmask) dmask[(((size_t)b * g.dg + d) * K + t) * L + l] = gm;
  }
}

// ---------------------------------------------------------------------------------------------- channels_last path --
// x [B,H,W,C], offset [B,Ho,Wo,dg*2K], mask [B,Ho,Wo,dg*K], col [B*L, K, C]: the layout the training path keeps its maps in.
// One wavefront per (output pixel, tap, deformable group): the 64 lanes own 4 consecutive channels each, so every neighbour
// read, the column write and (backward) the gradient atomics are contiguous 1-KiB wave accesses; the two offsets and the mask
// are wave-uniform scalars.  The GEMM around it (col [B*L, K*C] x weight [O, K*C]^T) reads / writes NHWC directly.
struct DeformTap {
  float w1, w2, w3, w4;          // bilinear weights of (hl,wl) (hl,wh) (hh,wl) (hh,wh); 0 where the neighbour is outside
  float lh, lw;
  int hl, wl;
  bool a1, a2, a3, a4, inside;
};

__device__ __forceinline__ DeformTap deform_tap(float h, float w, int H, int W) {
  DeformTap t;
  t.inside = h > -1.f && w > -1.f && h < (float)H && w < (float)W;
  t.hl = (int)floorf(h); t.wl = (int)floorf(w);
  t.lh = h - (float)t.hl; t.lw = w - (float)t.wl;
  const float uh = 1.f - t.lh, uw = 1.f - t.lw;
  t.a1 = t.inside && t.hl >= 0 && t.wl >= 0;
  t.a2 = t.inside && t.hl >= 0 && t.wl + 1 <= W - 1;
  t.a3 = t.inside && t.hl + 1 <= H - 1 && t.wl >= 0;
  t.a4 = t.inside && t.hl + 1 <= H - 1 && t.wl + 1 <= W - 1;
  t.w1 = uh * uw; t.w2 = uh * t.lw; t.w3 = t.lh * uw; t.w4 = t.lh * t.lw;
  return t;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 fma4(float a, const float4 v, const float4 acc) {
  return make_float4(fmaf(a, v.x, acc.x), fmaf(a, v.y, acc.y), fmaf(a, v.z, acc.z), fmaf(a, v.w, acc.w));
}

__global__ void __launch_bounds__(256)
    deform_im2col_cl_kernel(const float* __restrict__ x, const float* __restrict__ offset, const float* __restrict__ mask,
                            DeformGeom g, float* __restrict__ col) {
  const int L = g.Ho * g.Wo, K = g.kh * g.kw, cpg = g.C / g.dg;
  const int lane = threadIdx.x & 63;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  const long total = (long)g.B * L * K * g.dg;
  for (long it = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; it < total; it += nw) {
    const int d = (int)(it % g.dg);
    const long r = it / g.dg;
    const int t = (int)(r % K);
    const long bl = r / K;
    const int b = (int)(bl / L), l = (int)(bl - (long)b * L);
    const int ho = l / g.Wo, wo = l - ho * g.Wo, ki = t / g.kw, kj = t - ki * g.kw;
    const float* off = offset + (size_t)bl * (g.dg * 2 * K) + d * 2 * K + 2 * t;
    const float h = (float)(ho * g.sh - g.ph + ki * g.dh) + off[0];
    const float w = (float)(wo * g.sw - g.pw + kj * g.dw) + off[1];
    const float m = mask ? mask[(size_t)bl * (g.dg * K) + d * K + t] : 1.f;
    const DeformTap q = deform_tap(h, w, g.H, g.W);
    const float* xb = x + (size_t)b * g.H * g.W * g.C;
    float* cp = col + ((size_t)bl * K + t) * g.C + d * cpg;
    for (int c = lane * 4; c < cpg; c += 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int ch = d * cpg + c;
      if (q.a1) v = fma4(q.w1, ld4(xb + ((size_t)q.hl * g.W + q.wl) * g.C + ch), v);
      if (q.a2) v = fma4(q.w2, ld4(xb + ((size_t)q.hl * g.W + q.wl + 1) * g.C + ch), v);
      if (q.a3) v = fma4(q.w3, ld4(xb + ((size_t)(q.hl + 1) * g.W + q.wl) * g.C + ch), v);
      if (q.a4) v = fma4(q.w4, ld4(xb + ((size_t)(q.hl + 1) * g.W + q.wl + 1) * g.C + ch), v);
      *reinterpret_cast<float4*>(cp + c) = make_float4(v.x * m, v.y * m, v.z * m, v.w * m);
    }
  }
}

// the partition of grad_x between deform_col2im_gather_kernel (offsets within NEAR_R) and the atomics of
// deform_col2im_cl_kernel (the rest); see the gather kernel below
constexpr int DX_CH = 64;
constexpr float NEAR_R = 2.f;
constexpr int NEAR_E = 7;                               // 2 * NEAR_R + 3 displacements per axis

__device__ __forceinline__ bool near_sample(float oy, float ox) { return fabsf(oy) <= NEAR_R && fabsf(ox) <= NEAR_R; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Backward of the gather, all three gradients in one pass over grad_col: grad_x by contiguous f32 atomics (grad_x == NULL:
// skipped), grad_offset / grad_mask by a wave reduction over the channels of the deformable group.
__global__ void __launch_bounds__(256)
    deform_col2im_cl_kernel(const float* __restrict__ gcol, const float* __restrict__ x, const float* __restrict__ offset,
                            const float* __restrict__ mask, DeformGeom g, float* __restrict__ dx, int far_only,
                            float* __restrict__ doffset, float* __restrict__ dmask) {
  const int L = g.Ho * g.Wo, K = g.kh * g.kw, cpg = g.C / g.dg;
  const int lane = threadIdx.x & 63;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  const long total = (long)g.B * L * K * g.dg;
  for (long it = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; it < total; it += nw) {
    const int d = (int)(it % g.dg);
    const long r = it / g.dg;
    const int t = (int)(r % K);
    const long bl = r / K;
    const int b = (int)(bl / L), l = (int)(bl - (long)b * L);
    const int ho = l / g.Wo, wo = l - ho * g.Wo, ki = t / g.kw, kj = t - ki * g.kw;
    const size_t oi = (size_t)bl * (g.dg * 2 * K) + d * 2 * K + 2 * t;
    const float h = (float)(ho * g.sh - g.ph + ki * g.dh) + offset[oi];
    const float w = (float)(wo * g.sw - g.pw + kj * g.dw) + offset[oi + 1];
    const bool scatter = dx && !(far_only && near_sample(offset[oi], offset[oi + 1]));   // near samples: the gather kernel's
    const size_t mi = (size_t)bl * (g.dg * K) + d * K + t;
    const float m = mask ? mask[mi] : 1.f;
    const DeformTap q = deform_tap(h, w, g.H, g.W);
    const float uh = 1.f - q.lh, uw = 1.f - q.lw;
    const float* xb = x + (size_t)b * g.H * g.W * g.C;
    float* db = scatter ? dx + (size_t)b * g.H * g.W * g.C : nullptr;
    const float* gp = gcol + ((size_t)bl * K + t) * g.C + d * cpg;
    float gh = 0.f, gw = 0.f, gm = 0.f;
    if (q.inside) {
      for (int c = lane * 4; c < cpg; c += 256) {
        const int ch = d * cpg + c;
        const float4 gv = ld4(gp + c);
        const float gvs[4] = {gv.x, gv.y, gv.z, gv.w};
        const size_t p1 = ((size_t)q.hl * g.W + q.wl) * g.C + ch, p2 = p1 + g.C, p3 = p1 + (size_t)g.W * g.C, p4 = p3 + g.C;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v1 = q.a1 ? ld4(xb + p1) : z, v2 = q.a2 ? ld4(xb + p2) : z, v3 = q.a3 ? ld4(xb + p3) : z,
                     v4 = q.a4 ? ld4(xb + p4) : z;
        const float a1[4] = {v1.x, v1.y, v1.z, v1.w}, a2[4] = {v2.x, v2.y, v2.z, v2.w}, a3[4] = {v3.x, v3.y, v3.z, v3.w},
                    a4[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gj = gvs[j];
          gh += gj * m * (uw * (a3[j] - a1[j]) + q.lw * (a4[j] - a2[j]));
          gw += gj * m * (uh * (a2[j] - a1[j]) + q.lh * (a4[j] - a3[j]));
          gm += gj * (q.w1 * a1[j] + q.w2 * a2[j] + q.w3 * a3[j] + q.w4 * a4[j]);
          if (db) {
            const float gmj = gj * m;
            if (gmj != 0.f) {
              if (q.a1) atomicAdd(db + p1 + j, gmj * q.w1);
              if (q.a2) atomicAdd(db + p2 + j, gmj * q.w2);
              if (q.a3) atomicAdd(db + p3 + j, gmj * q.w3);
              if (q.a4) atomicAdd(db + p4 + j, gmj * q.w4);
            }
          }
        }
      }
    }
    gh = wave_sum(gh); gw = wave_sum(gw); gm = wave_sum(gm);
    if (lane == 0) {
      doffset[oi] = gh;
      doffset[oi + 1] = gw;
      if (dmask) dmask[mi] = gm;
    }
  }
}

// grad_x WITHOUT scatter for stride-1 convolutions.  Issued as atomics the 9 taps of a pixel and of its neighbours land on
// the same few input pixels and serialise (2.3 ms at the tower shape with global atomics, 1.1 ms staged through LDS float
// atomics, which run at ~0.4 lane-adds per clock and CU here).  Turned around: an input pixel (py, px) receives from the
// sample of (output pixel, tap) the weight hat(h - py) * hat(w - px), hat(u) = max(0, 1 - |u|), so it only has to look at the
// samples whose base position lies within NEAR_R + 1 pixels - as long as the learned offset itself is at most NEAR_R.
// One wavefront per (input pixel, 64-channel slice): the lanes evaluate the K * (2 NEAR_R + 3)^2 candidates 64 at a time
// (offsets are L2-resident), a ballot keeps the few with non-zero weight (~36), and for each of them the wave reads the 256
// contiguous bytes of its grad_col row.  Deterministic, no atomics.  Samples with an offset beyond NEAR_R are the
// complement: deform_col2im_cl_kernel adds exactly those with global atomics (`far_only`).
__global__ void __launch_bounds__(256)
    deform_col2im_gather_kernel(const float* __restrict__ gcol, const float* __restrict__ offset, const float* __restrict__ mask,
                                DeformGeom g, float* __restrict__ dx) {
  const int L = g.Ho * g.Wo, K = g.kh * g.kw, cpg = g.C / g.dg, nchunk = g.C / DX_CH;
  const int lane = threadIdx.x & 63;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wid >= (long)g.B * g.H * g.W * nchunk) return;                  // wave-uniform
  const int chunk = (int)(wid % nchunk);
  const long pix = wid / nchunk;
  const int px = (int)(pix % g.W), py = (int)((pix / g.W) % g.H), b = (int)(pix / ((long)g.W * g.H));
  const int ch0 = chunk * DX_CH, d = ch0 / cpg;
  const int ncand = K * NEAR_E * NEAR_E;
  float acc = 0.f;
  for (int base = 0; base < ncand; base += 64) {
    const int k = base + lane;
    float wgt = 0.f;
    int row = 0;                                                      // row of grad_col: (b * L + l) * K + t
    if (k < ncand) {
      const int t = k / (NEAR_E * NEAR_E), e = k - t * NEAR_E * NEAR_E, ey = e / NEAR_E - (NEAR_E / 2), ex = e % NEAR_E - (NEAR_E / 2);
      const int ki = t / g.kw, kj = t - ki * g.kw;
      const int ho = py + g.ph - ki * g.dh + ey, wo = px + g.pw - kj * g.dw + ex;      // stride 1: base position = (py + ey, px + ex)
      if (ho >= 0 && ho < g.Ho && wo >= 0 && wo < g.Wo) {
        const long bl = (long)b * L + ho * g.Wo + wo;
        const size_t oi = (size_t)bl * (g.dg * 2 * K) + d * 2 * K + 2 * t;
        const float oy = offset[oi], ox = offset[oi + 1];
        if (near_sample(oy, ox)) {
          const float h = (float)(ho - g.ph + ki * g.dh) + oy, w = (float)(wo - g.pw + kj * g.dw) + ox;
          const float wy = 1.f - fabsf(h - (float)py), wx = 1.f - fabsf(w - (float)px);
          if (wy > 0.f && wx > 0.f) {
            wgt = wy * wx * (mask ? mask[(size_t)bl * (g.dg * K) + d * K + t] : 1.f);
            row = (int)(bl * K + t);
          }
        }
      }
    }
    unsigned long long hits = __ballot(wgt != 0.f);
    while (hits) {
      const int j = __ffsll((long long)hits) - 1;
      hits &= hits - 1;
      const float wj = __shfl(wgt, j, 64);
      const int rj = __shfl(row, j, 64);
      acc = fmaf(wj, gcol[(size_t)rj * g.C + ch0 + lane], acc);
    }
  }
  dx[(size_t)pix * g.C + ch0 + lane] += acc;                         // the only writer of this element while it runs
}

}  // namespace pt

using namespace pt;

static int deform_geom(const char* fn, int B, int C, int H, int W, int kh, int kw, int ph, int pw, int sh, int sw, int dh,
                       int dw, int dg, DeformGeom* g) {
  PT_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && dg > 0 &&
                 ph >= 0 && pw >= 0,
             PT_EINVAL, "%s: bad size", fn);
  PT_REQUIRE(C % dg == 0, PT_EINVAL, "%s: C=%d not divisible by deform_groups=%d", fn, C, dg);
  const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1, Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  PT_REQUIRE(Ho > 0 && Wo > 0, PT_EINVAL, "%s: empty output", fn);
  *g = DeformGeom{B, C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, dg, Ho, Wo};
  return PT_OK;
}

static int blocks_for(long total) {
  long nb = (total + 255) / 256;
  return (int)(nb > 65536 ? 65536 : (nb < 1 ? 1 : nb));
}

extern "C" int pt_deform_im2col(const float* x, const float* offset, const float* mask, int B, int C, int H, int W,
                                int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                                int deform_groups, float* col, void* stream) {
  PT_REQUIRE(x && offset && col, PT_EINVAL, "pt_deform_im2col: NULL pointer");
  DeformGeom g;
  int rc = deform_geom("pt_deform_im2col", B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deform_groups, &g);
  if (rc) return rc;
  hipLaunchKernelGGL(deform_im2col_kernel, dim3(blocks_for((long)B * C * g.Ho * g.Wo)), dim3(256), 0, as_stream(stream), x,
                     offset, mask, g, col);
  PT_LAUNCH_CHECK("pt_deform_im2col");
  return PT_OK;
}

extern "C" int pt_deform_col2im(const float* grad_col, const float* offset, const float* mask, int B, int C, int H, int W,
                                int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                                int deform_groups, float* grad_x, void* stream) {
  PT_REQUIRE(grad_col && offset && grad_x, PT_EINVAL, "pt_deform_col2im: NULL pointer");
  DeformGeom g;
  int rc = deform_geom("pt_deform_col2im", B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deform_groups, &g);
  if (rc) return rc;
  hipLaunchKernelGGL(deform_col2im_kernel, dim3(blocks_for((long)B * C * kh * kw * g.Ho * g.Wo)), dim3(256), 0,
                     as_stream(stream), grad_col, offset, mask, g, grad_x);
  PT_LAUNCH_CHECK("pt_deform_col2im");
  return PT_OK;
}

extern "C" int pt_deform_col2im_coord(const float* grad_col, const float* x, const float* offset, const float* mask, int B,
                                      int C, int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                                      int dil_h, int dil_w, int deform_groups, float* grad_offset, float* grad_mask,
                                      void* stream) {
  PT_REQUIRE(grad_col && x && offset && grad_offset && (!grad_mask || mask), PT_EINVAL, "pt_deform_col2im_coord: NULL pointer");
  DeformGeom g;
  int rc = deform_geom("pt_deform_col2im_coord", B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
                       deform_groups, &g);
  if (rc) return rc;
  hipLaunchKernelGGL(deform_col2im_coord_kernel, dim3(blocks_for((long)B * deform_groups * kh * kw * g.Ho * g.Wo)), dim3(256),
                     0, as_stream(stream), grad_col, x, offset, mask, g, grad_offset, grad_mask);
  PT_LAUNCH_CHECK("pt_deform_col2im_coord");
  return PT_OK;
}

static int waves_grid(long items) {                       // 4 wavefronts per workgroup, one item per wavefront and trip
  long nb = (items + 3) / 4;
  return (int)(nb > 16384 ? 16384 : (nb < 1 ? 1 : nb));
}

extern "C" int pt_deform_im2col_cl(const float* x, const float* offset, const float* mask, int B, int C, int H, int W, int kh,
                                   int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                                   int deform_groups, float* col, void* stream) {
  PT_REQUIRE(x && offset && col, PT_EINVAL, "pt_deform_im2col_cl: NULL pointer");
  DeformGeom g;
  int rc = deform_geom("pt_deform_im2col_cl", B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deform_groups, &g);
  if (rc) return rc;
  PT_REQUIRE((C / deform_groups) % 4 == 0, PT_EINVAL, "pt_deform_im2col_cl: channels per deformable group (%d) must be a multiple of 4",
             C / deform_groups);
  hipLaunchKernelGGL(deform_im2col_cl_kernel, dim3(waves_grid((long)B * g.Ho * g.Wo * kh * kw * deform_groups)), dim3(256), 0,
                     as_stream(stream), x, offset, mask, g, col);
  PT_LAUNCH_CHECK("pt_deform_im2col_cl");
  return PT_OK;
}

extern "C" int pt_deform_col2im_cl(const float* grad_col, const float* x, const float* offset, const float* mask, int B, int C,
                                   int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                                   int dil_w, int deform_groups, float* grad_x, float* grad_offset, float* grad_mask,
                                   void* stream) {
  PT_REQUIRE(grad_col && x && offset && grad_offset && (!grad_mask || mask), PT_EINVAL, "pt_deform_col2im_cl: NULL pointer");
  DeformGeom g;
  int rc = deform_geom("pt_deform_col2im_cl", B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deform_groups, &g);
  if (rc) return rc;
  PT_REQUIRE((C / deform_groups) % 4 == 0, PT_EINVAL, "pt_deform_col2im_cl: channels per deformable group (%d) must be a multiple of 4",
             C / deform_groups);
  // stride 1 and 64-channel slices inside one deformable group: grad_x by the gather kernel, far samples by atomics
  const int cpg = C / deform_groups;
  const bool gather = grad_x && stride_h == 1 && stride_w == 1 && cpg % DX_CH == 0 && (long)B * g.Ho * g.Wo * kh * kw < 2147483647L;
  if (gather) {
    const long waves = (long)B * H * W * (C / DX_CH);
    PT_REQUIRE((waves + 3) / 4 <= 2147483647L, PT_ELIMIT, "pt_deform_col2im_cl: grid too large");
    hipLaunchKernelGGL(deform_col2im_gather_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream), grad_col,
                       offset, mask, g, grad_x);
    PT_LAUNCH_CHECK("pt_deform_col2im_cl(gather)");
  }
  hipLaunchKernelGGL(deform_col2im_cl_kernel, dim3(waves_grid((long)B * g.Ho * g.Wo * kh * kw * deform_groups)), dim3(256), 0,
                     as_stream(stream), grad_col, x, offset, mask, g, grad_x, gather ? 1 : 0, grad_offset, grad_mask);
  PT_LAUNCH_CHECK("pt_deform_col2im_cl");
  return PT_OK;
}
