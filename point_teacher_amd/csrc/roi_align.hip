// RoIAlign forward / backward for gfx950 (replaces mmcv.ops.RoIAlign at
// dense_heads/fcos_head_p2b_ts.py:1202,1243,1268; aligned=True, avg pooling,
// adaptive sampling grid ceil(roi/bin)).
//
// channels_last path (the one the training loop uses): one workgroup per RoI, one
// thread per channel.  All sample coordinates are wave-uniform, every neighbour read is a
// 256-float contiguous row of the [B,H,W,C] map (1 KiB coalesced, L2/MALL resident: the
// whole map is 20 MB), the 49 x C tile is transposed through LDS (row stride C+1 words,
// conflict-free) and stored as the contiguous 49*C block of out[K,C,7,7] that the FC stack
// flattens.  HBM traffic ~= the output bytes (K*C*49*4).  Backward mirrors it: the RoI's
// grad block is read contiguously into LDS and scattered with f32 atomics whose wave
// footprint is 256 contiguous bytes (the full-rate shape on gfx950).
#include "pt_common.h"

namespace pt {

struct RoiGeom {
  int b;
  float start_w, start_h, bin_w, bin_h;
  int grid_w, grid_h;
  float inv_count;
};

__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ roi, int out_size, float scale,
                                            int sampling_ratio, int aligned, int B) {
  RoiGeom g;
  g.b = min(max((int)roi[0], 0), B - 1);  // never index outside the batch
  const float off = aligned ? 0.5f : 0.f;
  g.start_w = roi[1] * scale - off;
  g.start_h = roi[2] * scale - off;
  const float end_w = roi[3] * scale - off, end_h = roi[4] * scale - off;
  float rw = end_w - g.start_w, rh = end_h - g.start_h;
  if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  g.bin_h = rh / (float)out_size;
  g.bin_w = rw / (float)out_size;
  g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)out_size);
  g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)out_size);
  const float cnt = fmaxf((float)(g.grid_h * g.grid_w), 1.f);
  g.inv_count = cnt;  // keep the divisor: results are divided, as the reference does
  return g;
}

struct Bilin {
  int y0, y1, x0, x1;
  float w1, w2, w3, w4;
  bool valid;
};

__device__ __forceinline__ Bilin bilin(float y, float x, int H, int W) {
  Bilin r;
  r.valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
  const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
  r.y0 = yl; r.y1 = yh; r.x0 = xl; r.x1 = xh;
  r.w1 = hy * hx; r.w2 = hy * lx; r.w3 = ly * hx; r.w4 = ly * lx;
  return r;
}

constexpr int MAX_BINS = 49;  // out_size <= 7

// ------------------------------------------------------------ channels_last --
// Separable form.  The samples of a bin lie on a product grid ys x xs and bilinear
// interpolation (and the "outside the map" test) is separable, so
//     out[ph][pw][c] = 1/count * sum_py sum_px  Ay[ph][py] * Ax[pw][px] * feat[y0+py][x0+px][c]
// with per-axis weight vectors Ay[ph][.] / Ax[pw][.] that are non-zero only on the band of
// pixel rows / columns bin ph / pw touches.  A bin with a g x g sampling grid needs (g+1)^2
// pixel reads instead of 4 g^2 taps, the weights are computed once per RoI (not per channel)
// and every index is wave-uniform.  AxisW lives in LDS.
constexpr int MAXR = 256;   // largest supported map extent (H, W <= 256)
constexpr int NB = 7;       // out_size <= 7

struct AxisW {
  float w[NB][MAXR];  // dense [bin][pixel - o]; zero outside the band
  int lo[NB], hi[NB];  // inclusive band of each bin (hi < lo: empty)
  int o, e;            // first / last pixel touched by any bin (e < o: nothing)
};

// Block-cooperative.  Mirrors bilin(): v <= 0 -> 0; lo >= L-1 -> lo = hi = L-1, frac 0.
__device__ void axis_weights(float start, float bin, int grid, int L, int out_size, AxisW* A) {
  for (int i = threadIdx.x; i < NB * MAXR; i += blockDim.x) (&A->w[0][0])[i] = 0.f;
  if (threadIdx.x < NB) { A->lo[threadIdx.x] = 1 << 30; A->hi[threadIdx.x] = -1; }
  if (threadIdx.x == 0) { A->o = 1 << 30; A->e = -1; }
  __syncthreads();
  const int n = out_size * grid;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {   // pass 1: extent
    const int p = t / grid, i = t - p * grid;
    float v = start + p * bin + (i + .5f) * bin / (float)grid;
    if (v < -1.0f || v > (float)L) continue;
    if (v <= 0.f) v = 0.f;
    int l = (int)v, h;
    if (l >= L - 1) { h = l = L - 1; } else { h = l + 1; }
    atomicMin(&A->o, l);
    atomicMax(&A->e, h);
  }
  __syncthreads();
  const int o = A->o;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {   // pass 2: weights and bands
    const int p = t / grid, i = t - p * grid;
    float v = start + p * bin + (i + .5f) * bin / (float)grid;
    if (v < -1.0f || v > (float)L) continue;
    if (v <= 0.f) v = 0.f;
    int l = (int)v, h;
    if (l >= L - 1) { h = l = L - 1; v = (float)l; } else { h = l + 1; }
    const float fl = v - (float)l, fh = 1.f - fl;
    atomicAdd(&A->w[p][l - o], fh);
    atomicAdd(&A->w[p][h - o], fl);
    atomicMin(&A->lo[p], l - o);
    atomicMax(&A->hi[p], h - o);
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256)
    roi_align_fwd_cl(const float* __restrict__ feat, const float* __restrict__ rois, int B, int C, int H, int W,
                     int out_size, float scale, int sampling_ratio, int aligned, float* __restrict__ out,
                     const int* __restrict__ fallback, int group) {
  extern __shared__ float tile[];  // [bins][C+1]
  __shared__ AxisW AX, AY;
  const int k = blockIdx.x;
  if (fallback && !fallback[k]) return;   // this RoI was done by the small-footprint kernel
  const RoiGeom g = roi_geom(rois + (size_t)k * 5, out_size, scale, sampling_ratio, aligned, B);
  const int bins = out_size * out_size;
  const int ld = C + 1;
  for (int i = threadIdx.x; i < bins * ld; i += blockDim.x) tile[i] = 0.f;
  axis_weights(g.start_w, g.bin_w, g.grid_w, W, out_size, &AX);
  axis_weights(g.start_h, g.bin_h, g.grid_h, H, out_size, &AY);
  const float* fb = feat + (size_t)g.b * H * W * C;
  const int ny = AY.e - AY.o + 1;
  if (AX.e >= AX.o && ny > 0) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      for (int py = 0; py < ny; ++py) {
        const float* row = fb + ((size_t)(AY.o + py) * W + AX.o) * C + c;
        for (int pw = 0; pw < out_size; ++pw) {
          const int xl = AX.lo[pw], xh = AX.hi[pw];
          if (xh < xl) continue;
          float t = 0.f;
          for (int px = xl; px <= xh; ++px) t += AX.w[pw][px] * row[(size_t)px * C];
          for (int ph = 0; ph < out_size; ++ph)
            if (AY.lo[ph] <= py && py <= AY.hi[ph]) tile[(ph * out_size + pw) * ld + c] += AY.w[ph][py] * t;
        }
      }
    }
  }
  __syncthreads();
  float* ob = out + (size_t)k * C * bins;
  for (int o = threadIdx.x; o < C * bins; o += blockDim.x) {
    const int c = o / bins, bin = o - c * bins;
    ob[o] = tile[bin * ld + c] / g.inv_count;
  }
}

// Both axes at once, zeroing only the columns the RoI touches (3 barriers, ~7*(nx+ny) LDS words cleared
// instead of 2*7*256).  Same arithmetic as axis_weights().
__device__ void axis_weights2(float sx, float bx, int gx, int Lx, float sy, float by, int gy, int Ly, int out_size,
                              AxisW* AX, AxisW* AY) {
  if (threadIdx.x < NB) { AX->lo[threadIdx.x] = 1 << 30; AX->hi[threadIdx.x] = -1; }
  else if (threadIdx.x < 2 * NB) { AY->lo[threadIdx.x - NB] = 1 << 30; AY->hi[threadIdx.x - NB] = -1; }
  if (threadIdx.x == 32) { AX->o = 1 << 30; AX->e = -1; AY->o = 1 << 30; AY->e = -1; }
  __syncthreads();
  const int nxs = out_size * gx, nys = out_size * gy;
  for (int t = threadIdx.x; t < nxs + nys; t += blockDim.x) {   // pass 1: extents
    const bool isx = t < nxs;
    const int u = isx ? t : t - nxs, grid = isx ? gx : gy, L = isx ? Lx : Ly;
    const float start = isx ? sx : sy, bin = isx ? bx : by;
    AxisW* A = isx ? AX : AY;
    const int p = u / grid, i = u - p * grid;
    float v = start + p * bin + (i + .5f) * bin / (float)grid;
    if (v < -1.0f || v > (float)L) continue;
    if (v <= 0.f) v = 0.f;
    int l = (int)v, h;
    if (l >= L - 1) { h = l = L - 1; } else { h = l + 1; }
    atomicMin(&A->o, l);
    atomicMax(&A->e, h);
  }
  __syncthreads();
  const int ox = AX->o, oy = AY->o;
  const int nx = max(AX->e - ox + 1, 0), ny = max(AY->e - oy + 1, 0);
  for (int t = threadIdx.x; t < NB * (nx + ny); t += blockDim.x) {   // clear only what is used
    if (t < NB * nx) AX->w[t / nx][t % nx] = 0.f;
    else { const int u = t - NB * nx; AY->w[u / ny][u % ny] = 0.f; }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nxs + nys; t += blockDim.x) {   // pass 2: weights and bands
    const bool isx = t < nxs;
    const int u = isx ? t : t - nxs, grid = isx ? gx : gy, L = isx ? Lx : Ly, o = isx ? ox : oy;
    const float start = isx ? sx : sy, bin = isx ? bx : by;
    AxisW* A = isx ? AX : AY;
    const int p = u / grid, i = u - p * grid;
    float v = start + p * bin + (i + .5f) * bin / (float)grid;
    if (v < -1.0f || v > (float)L) continue;
    if (v <= 0.f) v = 0.f;
    int l = (int)v, h;
    if (l >= L - 1) { h = l = L - 1; v = (float)l; } else { h = l + 1; }
    const float fl = v - (float)l, fh = 1.f - fl;
    atomicAdd(&A->w[p][l - o], fh);
    atomicAdd(&A->w[p][h - o], fl);
    atomicMin(&A->lo[p], l - o);
    atomicMax(&A->hi[p], h - o);
  }
  __syncthreads();
}

// Forward for out_size == 7 (every config): the 49 bin sums of channel c live in 49 REGISTERS of thread c.
// Per footprint pixel: one coalesced global read + 7 FMAs against the dense per-bin column weights (LDS
// broadcast), per footprint row: 49 FMAs against the row weights.  The LDS tile is written once, for the
// transpose to the [C][49] output order.
__global__ void __launch_bounds__(256)
    roi_align_fwd_cl7(const float* __restrict__ feat, const float* __restrict__ rois, int B, int C, int H, int W,
                      float scale, int sampling_ratio, int aligned, float* __restrict__ out,
                      const int* __restrict__ fallback) {
  extern __shared__ float tile[];  // [49][C+1]
  __shared__ AxisW AX, AY;
  const int k = blockIdx.x;
  if (fallback && !fallback[k]) return;   // this RoI was done by the small-footprint kernel
  const RoiGeom g = roi_geom(rois + (size_t)k * 5, 7, scale, sampling_ratio, aligned, B);
  const int ld = C + 1;
  axis_weights2(g.start_w, g.bin_w, g.grid_w, W, g.start_h, g.bin_h, g.grid_h, H, 7, &AX, &AY);
  const float* fb = feat + (size_t)g.b * H * W * C;
  const int ny = AY.e - AY.o + 1, nx = AX.e - AX.o + 1;
  const float inv = 1.f / g.inv_count;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc[7][7];
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
      for (int b = 0; b < 7; ++b) acc[a][b] = 0.f;
    if (nx > 0 && ny > 0) {
      for (int py = 0; py < ny; ++py) {
        const float* row = fb + ((size_t)(AY.o + py) * W + AX.o) * C + c;
        float t[7];
#pragma unroll
        for (int b = 0; b < 7; ++b) t[b] = 0.f;
        for (int px = 0; px < nx; ++px) {
          const float v = row[(size_t)px * C];
#pragma unroll
          for (int b = 0; b < 7; ++b) t[b] += AX.w[b][px] * v;
        }
#pragma unroll
        for (int a = 0; a < 7; ++a) {
          const float wy = AY.w[a][py];
#pragma unroll
          for (int b = 0; b < 7; ++b) acc[a][b] += wy * t[b];
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
      for (int b = 0; b < 7; ++b) tile[(a * 7 + b) * ld + c] = acc[a][b] / g.inv_count;
  }
  (void)inv;
  __syncthreads();
  float* ob = out + (size_t)k * C * 49;
  for (int o = threadIdx.x; o < C * 49; o += blockDim.x) {
    const int c = o / 49, bin = o - c * 49;
    ob[o] = tile[bin * ld + c];
  }
}

// Backward, channels_last, generic: one workgroup per RoI; with the separable weights a RoI issues
// ONE f32 atomic per footprint pixel and channel (256 contiguous bytes per wave instruction - the
// full-rate shape) instead of 4 per bilinear tap.  RoIs whose bag was reduced on chip by the
// small-footprint kernel below are skipped through `fallback`.
__global__ void __launch_bounds__(256)
    roi_align_bwd_cl(const float* __restrict__ gout, const float* __restrict__ rois, int B, int C, int H, int W, int K,
                     int group, int out_size, float scale, int sampling_ratio, int aligned,
                     float* __restrict__ gfeat, const int* __restrict__ fallback) {
  extern __shared__ float tile[];  // [bins][C+1] grad tile
  __shared__ AxisW AX, AY;
  const int k = blockIdx.x;
  if (fallback && !fallback[k / group]) return;   // done by the small-footprint kernel
  const int bins = out_size * out_size;
  const int ld = C + 1;
  const RoiGeom g = roi_geom(rois + (size_t)k * 5, out_size, scale, sampling_ratio, aligned, B);
  const float* gb = gout + (size_t)k * C * bins;
  for (int o = threadIdx.x; o < C * bins; o += blockDim.x) {
    const int c = o / bins, bin = o - c * bins;
    tile[bin * ld + c] = gb[o];
  }
  axis_weights2(g.start_w, g.bin_w, g.grid_w, W, g.start_h, g.bin_h, g.grid_h, H, out_size, &AX, &AY);   // ends with a barrier
  const int ny = AY.e - AY.o + 1, nx = AX.e - AX.o + 1;
  if (ny <= 0 || nx <= 0) return;
  float* fb = gfeat + (size_t)g.b * H * W * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    for (int py = 0; py < ny; ++py) {
      float S[NB];
#pragma unroll
      for (int pw = 0; pw < NB; ++pw) S[pw] = 0.f;
      for (int ph = 0; ph < out_size; ++ph) {
        if (AY.lo[ph] > py || py > AY.hi[ph]) continue;
        const float wy = AY.w[ph][py] / g.inv_count;
#pragma unroll
        for (int pw = 0; pw < NB; ++pw)
          if (pw < out_size) S[pw] += wy * tile[(ph * out_size + pw) * ld + c];
      }
      for (int px = 0; px < nx; ++px) {
        float v = 0.f;
#pragma unroll
        for (int pw = 0; pw < NB; ++pw)
          if (pw < out_size) v += AX.w[pw][px] * S[pw];
        if (v != 0.f) atomicAdd(&fb[((size_t)(AY.o + py) * W + AX.o + px) * C + c], v);
      }
    }
  }
}

// ------------------------------------------------ small-footprint bag fast path --
// The U2 jittered boxes of one MIL bag of a tiny object (the common case: AI-TOD objects are
// ~12 px = 1.5 feature pixels) all fall on the same <= 5x5 feature pixels.  One workgroup owns
// the bag: the 25 footprint pixels of channel c live in 25 REGISTERS of thread c (loaded once
// for the whole bag), the separable per-axis weights of every member are precomputed into LDS
// by the first threads (one thread per (member, axis, bin): no atomics), and each member costs
// 420 register FMAs per thread - no barriers, no LDS tile, no per-tap memory traffic.  The
// backward keeps the 25 footprint accumulators in registers across the whole bag and issues 25
// coalesced atomics per thread at the end.  A workgroup falls back to the generic kernels
// (flag in `fallback[blockIdx]`) when its union footprint is larger or spans two images.
constexpr int SF = 5;            // footprint side
constexpr int SMALL_GROUP = 32;  // members per workgroup (>= U1*U2 of the 0 % config)

struct SmallW {
  float ax[SMALL_GROUP][NB][SF];
  float ay[SMALL_GROUP][NB][SF];   // already divided by the sample count
  int ub[6];                       // x0, y0, x1, y1, batch, ok
};

__device__ __forceinline__ void small_tap(float v, int L, int& l, int& h, float& fl, bool& valid) {
  valid = !(v < -1.0f || v > (float)L);
  if (v <= 0.f) v = 0.f;
  l = (int)v;
  if (l >= L - 1) { h = l = L - 1; v = (float)l; } else { h = l + 1; }
  fl = v - (float)l;
}

// Fills S for RoIs [k0,k1).  Returns (via S->ub[5]) whether the fast path applies.
__device__ void small_setup(const float* __restrict__ rois, int k0, int k1, int B, int H, int W, int out_size,
                            float scale, int sampling_ratio, int aligned, SmallW* S) {
  if (threadIdx.x == 0) { S->ub[0] = 1 << 30; S->ub[1] = 1 << 30; S->ub[2] = -1; S->ub[3] = -1; S->ub[4] = -1; S->ub[5] = 1; }
  __syncthreads();
  const int n = (k1 - k0) * 2 * out_size;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {      // pass 1: exact union of the taps
    const int r = t / (2 * out_size), a = (t / out_size) & 1, p = t % out_size;
    const RoiGeom g = roi_geom(rois + (size_t)(k0 + r) * 5, out_size, scale, sampling_ratio, aligned, B);
    const float start = a ? g.start_h : g.start_w, bin = a ? g.bin_h : g.bin_w;
    const int grid = a ? g.grid_h : g.grid_w, L = a ? H : W;
    if (grid > 8) S->ub[5] = 0;
    for (int i = 0; i < grid && i < 8; ++i) {
      int l, h; float fl; bool valid;
      small_tap(start + p * bin + (i + .5f) * bin / (float)grid, L, l, h, fl, valid);
      if (!valid) continue;
      atomicMin(&S->ub[a], l);
      atomicMax(&S->ub[2 + a], h);
    }
    if (a == 0 && p == 0) {
      const int old = atomicCAS(&S->ub[4], -1, g.b);
      if (old != -1 && old != g.b) S->ub[5] = 0;
    }
  }
  __syncthreads();
  const int ox = S->ub[0], oy = S->ub[1];
  const bool ok = S->ub[5] && (S->ub[2] - ox < SF) && (S->ub[3] - oy < SF);
  __syncthreads();
  if (threadIdx.x == 0) S->ub[5] = ok ? 1 : 0;
  if (ok) {
    for (int t = threadIdx.x; t < n; t += blockDim.x) {    // pass 2: one thread per (member, axis, bin)
      const int r = t / (2 * out_size), a = (t / out_size) & 1, p = t % out_size;
      const RoiGeom g = roi_geom(rois + (size_t)(k0 + r) * 5, out_size, scale, sampling_ratio, aligned, B);
      const float start = a ? g.start_h : g.start_w, bin = a ? g.bin_h : g.bin_w;
      const int grid = a ? g.grid_h : g.grid_w, L = a ? H : W, o = a ? oy : ox;
      float* w = a ? S->ay[r][p] : S->ax[r][p];
#pragma unroll
      for (int i = 0; i < SF; ++i) w[i] = 0.f;
      for (int i = 0; i < grid; ++i) {
        int l, h; float fl; bool valid;
        small_tap(start + p * bin + (i + .5f) * bin / (float)grid, L, l, h, fl, valid);
        if (!valid) continue;
        w[l - o] += 1.f - fl;
        w[h - o] += fl;
      }
      if (a) {
#pragma unroll
        for (int i = 0; i < SF; ++i) w[i] = w[i] / g.inv_count;
      }
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256)
    roi_align_small_fwd(const float* __restrict__ feat, const float* __restrict__ rois, int B, int C, int H, int W,
                        int K, int group, int out_size, float scale, int sampling_ratio, int aligned,
                        float* __restrict__ out, int* __restrict__ fallback) {
  __shared__ SmallW S;
  group = 1;   // the forward has nothing to share between bag members: one RoI per workgroup for parallelism
  const int k0 = blockIdx.x * group, k1 = min(k0 + group, K);
  small_setup(rois, k0, k1, B, H, W, out_size, scale, sampling_ratio, aligned, &S);
  if (!S.ub[5]) { if (threadIdx.x == 0) fallback[blockIdx.x] = 1; return; }
  if (threadIdx.x == 0) fallback[blockIdx.x] = 0;
  const int ox = S.ub[0], oy = S.ub[1], b = S.ub[4];
  const int bins = out_size * out_size;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float f[SF][SF];
#pragma unroll
    for (int y = 0; y < SF; ++y)
#pragma unroll
      for (int x = 0; x < SF; ++x) {
        const int yy = min(oy + y, H - 1), xx = min(ox + x, W - 1);   // rows/cols beyond the union have zero weight
        f[y][x] = feat[(((size_t)b * H + yy) * W + xx) * C + c];
      }
    for (int k = k0; k < k1; ++k) {
      const int r = k - k0;
      float* ob = out + ((size_t)k * C + c) * bins;
      float T[SF][NB];
#pragma unroll
      for (int y = 0; y < SF; ++y)
#pragma unroll
        for (int pw = 0; pw < NB; ++pw) {
          float t = 0.f;
          if (pw < out_size) {
#pragma unroll
            for (int x = 0; x < SF; ++x) t = fmaf(S.ax[r][pw][x], f[y][x], t);
          }
          T[y][pw] = t;
        }
#pragma unroll
      for (int ph = 0; ph < NB; ++ph) {
        if (ph < out_size) {
#pragma unroll
          for (int pw = 0; pw < NB; ++pw) {
            if (pw < out_size) {
              float v = 0.f;
#pragma unroll
              for (int y = 0; y < SF; ++y) v = fmaf(S.ay[r][ph][y], T[y][pw], v);
              ob[ph * out_size + pw] = v;
            }
          }
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256)
    roi_align_small_bwd(const float* __restrict__ gout, const float* __restrict__ rois, int B, int C, int H, int W,
                        int K, int group, int out_size, float scale, int sampling_ratio, int aligned,
                        float* __restrict__ gfeat, int* __restrict__ fallback) {
  __shared__ SmallW S;
  const int k0 = blockIdx.x * group, k1 = min(k0 + group, K);
  small_setup(rois, k0, k1, B, H, W, out_size, scale, sampling_ratio, aligned, &S);
  if (!S.ub[5]) { if (threadIdx.x == 0) fallback[blockIdx.x] = 1; return; }
  if (threadIdx.x == 0) fallback[blockIdx.x] = 0;
  const int ox = S.ub[0], oy = S.ub[1], b = S.ub[4];
  const int bins = out_size * out_size;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc[SF][SF];
#pragma unroll
    for (int y = 0; y < SF; ++y)
#pragma unroll
      for (int x = 0; x < SF; ++x) acc[y][x] = 0.f;
    for (int k = k0; k < k1; ++k) {
      const int r = k - k0;
      const float* gb = gout + ((size_t)k * C + c) * bins;
      float Sy[SF][NB];
#pragma unroll
      for (int y = 0; y < SF; ++y)
#pragma unroll
        for (int pw = 0; pw < NB; ++pw) Sy[y][pw] = 0.f;
#pragma unroll
      for (int ph = 0; ph < NB; ++ph) {
        if (ph < out_size) {
#pragma unroll
          for (int pw = 0; pw < NB; ++pw) {
            if (pw < out_size) {
              const float g = gb[ph * out_size + pw];
#pragma unroll
              for (int y = 0; y < SF; ++y) Sy[y][pw] = fmaf(S.ay[r][ph][y], g, Sy[y][pw]);
            }
          }
        }
      }
#pragma unroll
      for (int y = 0; y < SF; ++y)
#pragma unroll
        for (int x = 0; x < SF; ++x) {
          float v = acc[y][x];
#pragma unroll
          for (int pw = 0; pw < NB; ++pw)
            if (pw < out_size) v = fmaf(S.ax[r][pw][x], Sy[y][pw], v);
          acc[y][x] = v;
        }
    }
#pragma unroll
    for (int y = 0; y < SF; ++y)
#pragma unroll
      for (int x = 0; x < SF; ++x) {
        const float v = acc[y][x];
        if (v != 0.f && oy + y < H && ox + x < W)
          atomicAdd(&gfeat[(((size_t)b * H + oy + y) * W + ox + x) * C + c], v);
      }
  }
}

// ---------------------------------------------------------------- NCHW path --
__global__ void __launch_bounds__(256)
    roi_align_fwd_nchw(const float* __restrict__ feat, const float* __restrict__ rois, int B, long total, int C, int H,
                       int W, int out_size, float scale, int sampling_ratio, int aligned, float* __restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(i % out_size), ph = (int)((i / out_size) % out_size);
    const int c = (int)((i / (out_size * out_size)) % C);
    const int k = (int)(i / ((long)out_size * out_size * C));
    const RoiGeom g = roi_geom(rois + (size_t)k * 5, out_size, scale, sampling_ratio, aligned, B);
    const float* fb = feat + ((size_t)g.b * C + c) * H * W;
    float acc = 0.f;
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const float y = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.grid_h;
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const float x = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.grid_w;
        const Bilin q = bilin(y, x, H, W);
        if (q.valid)
          acc += q.w1 * fb[q.y0 * W + q.x0] + q.w2 * fb[q.y0 * W + q.x1] + q.w3 * fb[q.y1 * W + q.x0] +
                 q.w4 * fb[q.y1 * W + q.x1];
      }
    }
    out[i] = acc / g.inv_count;
  }
}

__global__ void __launch_bounds__(256)
    roi_align_bwd_nchw(const float* __restrict__ gout, const float* __restrict__ rois, int B, long total, int C, int H,
                       int W, int out_size, float scale, int sampling_ratio, int aligned, float* __restrict__ gfeat) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(i % out_size), ph = (int)((i / out_size) % out_size);
    const int c = (int)((i / (out_size * out_size)) % C);
    const int k = (int)(i / ((long)out_size * out_size * C));
    const RoiGeom g = roi_geom(rois + (size_t)k * 5, out_size, scale, sampling_ratio, aligned, B);
    float* fb = gfeat + ((size_t)g.b * C + c) * H * W;
    const float gv = gout[i] / g.inv_count;
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const float y = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.grid_h;
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const float x = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.grid_w;
        const Bilin q = bilin(y, x, H, W);
        if (q.valid) {
          atomicAdd(&fb[q.y0 * W + q.x0], gv * q.w1);
          atomicAdd(&fb[q.y0 * W + q.x1], gv * q.w2);
          atomicAdd(&fb[q.y1 * W + q.x0], gv * q.w3);
          atomicAdd(&fb[q.y1 * W + q.x1], gv * q.w4);
        }
      }
    }
  }
}

}  // namespace pt

using namespace pt;

static int roi_check(const char* fn, const void* a, const void* rois, const void* o, int B, int C, int H, int W,
                     int K, int out_size, int channels_last) {
  PT_REQUIRE(a && rois && o, PT_EINVAL, "%s: NULL pointer", fn);
  PT_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && K > 0, PT_EINVAL, "%s: bad size", fn);
  PT_REQUIRE(out_size >= 1 && out_size * out_size <= MAX_BINS, PT_ELIMIT, "%s: out_size=%d above 7", fn, out_size);
  if (channels_last) {
    PT_REQUIRE((size_t)(C + 1) * out_size * out_size * 4 + 2 * sizeof(pt::AxisW) <= 150 * 1024, PT_ELIMIT,
               "%s: C=%d too large for the LDS tile", fn, C);
    PT_REQUIRE(H <= pt::MAXR && W <= pt::MAXR, PT_ELIMIT, "%s: channels_last path supports maps up to %dx%d", fn,
               pt::MAXR, pt::MAXR);
  }
  return PT_OK;
}

extern "C" int pt_roi_align_fwd(const float* feat, const float* rois, int B, int C, int H, int W, int K, int out_size,
                                float spatial_scale, int sampling_ratio, int aligned, int channels_last, int group,
                                int32_t* group_ws, float* out, void* stream) {
  if (K == 0) return PT_OK;
  int rc = roi_check("pt_roi_align_fwd", feat, rois, out, B, C, H, W, K, out_size, channels_last);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  if (channels_last) {
    const size_t lds = (size_t)(C + 1) * out_size * out_size * sizeof(float);
    static size_t attr_bytes = 0;
    if (lds > attr_bytes) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_cl),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("pt_roi_align_fwd: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
      attr_bytes = lds;
    }
    const int* fb = nullptr;
    if (group < 1) group = 1;
    if (group_ws && out_size <= NB) {
      hipLaunchKernelGGL(roi_align_small_fwd, dim3(K), dim3(256), 0, s, feat, rois, B, C, H, W, K, 1,
                         out_size, spatial_scale, sampling_ratio, aligned, out, group_ws);
      PT_LAUNCH_CHECK("pt_roi_align_fwd(small)");
      fb = group_ws;
    }
    if (out_size == 7) {
      static size_t attr7 = 0;
      if (lds > attr7) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_cl7),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("pt_roi_align_fwd: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
        attr7 = lds;
      }
      hipLaunchKernelGGL(roi_align_fwd_cl7, dim3(K), dim3(256), lds, s, feat, rois, B, C, H, W, spatial_scale,
                         sampling_ratio, aligned, out, fb);
    } else {
      hipLaunchKernelGGL(roi_align_fwd_cl, dim3(K), dim3(256), lds, s, feat, rois, B, C, H, W, out_size, spatial_scale,
                         sampling_ratio, aligned, out, fb, group);
    }
  } else {
    const long total = (long)K * C * out_size * out_size;
    int nb = cdiv(total, 256);
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(roi_align_fwd_nchw, dim3(nb), dim3(256), 0, s, feat, rois, B, total, C, H, W, out_size,
                       spatial_scale, sampling_ratio, aligned, out);
  }
  PT_LAUNCH_CHECK("pt_roi_align_fwd");
  return PT_OK;
}

extern "C" int pt_roi_align_bwd(const float* grad_out, const float* rois, int B, int C, int H, int W, int K,
                                int out_size, float spatial_scale, int sampling_ratio, int aligned,
                                int channels_last, int group, int32_t* group_ws, float* grad_feat, void* stream) {
  if (K == 0) return PT_OK;
  int rc = roi_check("pt_roi_align_bwd", grad_out, rois, grad_feat, B, C, H, W, K, out_size, channels_last);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  if (channels_last) {
    const size_t lds = (size_t)(C + 1) * out_size * out_size * sizeof(float);
    PT_REQUIRE(lds + 2 * sizeof(AxisW) <= 160 * 1024, PT_ELIMIT, "pt_roi_align_bwd: C=%d too large for the LDS tiles", C);
    static size_t attr_bytes = 0;
    if (lds > attr_bytes) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_bwd_cl),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("pt_roi_align_bwd: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
      attr_bytes = lds;
    }
    if (group < 1) group = 1;
    if (group > 64) group = 64;
    const int* fb = nullptr;
    if (group_ws && group > 1 && group <= SMALL_GROUP && out_size <= NB) {
      hipLaunchKernelGGL(roi_align_small_bwd, dim3(cdiv(K, group)), dim3(256), 0, s, grad_out, rois, B, C, H, W, K,
                         group, out_size, spatial_scale, sampling_ratio, aligned, grad_feat, group_ws);
      PT_LAUNCH_CHECK("pt_roi_align_bwd(small)");
      fb = group_ws;
    }
    hipLaunchKernelGGL(roi_align_bwd_cl, dim3(K), dim3(256), lds, s, grad_out, rois, B, C, H, W, K, group,
                       out_size, spatial_scale, sampling_ratio, aligned, grad_feat, fb);
  } else {
    const long total = (long)K * C * out_size * out_size;
    int nb = cdiv(total, 256);
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(roi_align_bwd_nchw, dim3(nb), dim3(256), 0, s, grad_out, rois, B, total, C, H, W, out_size,
                       spatial_scale, sampling_ratio, aligned, grad_feat);
  }
  PT_LAUNCH_CHECK("pt_roi_align_bwd");
  return PT_OK;
}
