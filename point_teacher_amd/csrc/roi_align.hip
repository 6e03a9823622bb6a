// RoIAlign forward / backward for gfx950 (replaces mmcv.ops.RoIAlign at
// dense_heads/fcos_head_p2b_ts.py:1202,1243,1268; aligned=True, avg pooling,
// adaptive sampling grid ceil(roi/bin)).
//
// channels_last path (the one the training loop uses), out_size 7 (every config): ONE launch per
// direction (roi_align7_fwd / roi_align7_bwd).  A workgroup takes a run of consecutive RoIs of one
// MIL bag (they overlap), one thread per channel; every neighbour read is a 256-float contiguous
// row of the [B,H,W,C] map (1 KiB coalesced per wave), all sample coordinates are wave-uniform.
// The [C][49] block of a RoI is transposed through LDS (lane stride 49 words: conflict-free) and
// moved with 16-byte-per-lane fully coalesced stores / loads - it is the contiguous 49*C block of
// out[K,C,7,7] that the FC stack flattens, so HBM traffic ~= those bytes (K*C*49*4).  Three paths,
// chosen per workgroup / per RoI inside the launch:
//   A  the run's taps fall on <= 5x5 feature pixels (a bag of a tiny object - the common case): the
//      footprint of channel c lives in 25 registers for the whole run; the backward keeps 25
//      accumulators across the run and issues 25 coalesced atomics per thread at the end;
//   B  any RoI up to 104x104 feature pixels: separable per-axis weights (one thread per (axis, bin), no
//      atomics) in LDS, (g+1)^2 pixel reads per bin instead of 4 g^2 taps; the backward issues ONE f32
//      atomic per footprint pixel and channel (256 contiguous bytes per wave instruction);
//   R  (backward only) a run of 2..5 members whose own extents are <= 48 pixels per axis: the members' gradient blocks
//      (a 64-channel slice each) and all their axis weights sit in LDS together, a wavefront walks the rows of the UNION of
//      the footprints and adds the members' contributions in registers before the one atomic per union pixel - the 25
//      members of a bag are concentric, so a run's union is barely larger than its largest member (phase 1's synthetic
//      boxes: 3.9x fewer atomic adds at run length 5, profiles/r03/roi_stats_step1.txt);
//   C  larger RoIs: direct per-sample taps (correct, slow, never seen in training).
// Other out_sizes keep the generic one-workgroup-per-RoI kernels (roi_align_fwd_cl / _bwd_cl).
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
                     int out_size, float scale, int sampling_ratio, int aligned, float* __restrict__ out) {
  extern __shared__ float tile[];  // [bins][C+1]
  __shared__ AxisW AX, AY;
  const int k = blockIdx.x;
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

// Backward, channels_last, generic: one workgroup per RoI; with the separable weights a RoI issues
// ONE f32 atomic per footprint pixel and channel (256 contiguous bytes per wave instruction - the
// full-rate shape) instead of 4 per bilinear tap.  RoIs whose bag was reduced on chip by the
// small-footprint kernel below are skipped through `fallback`.
__global__ void __launch_bounds__(256)
    roi_align_bwd_cl(const float* __restrict__ gout, const float* __restrict__ rois, int B, int C, int H, int W, int K,
                     int group, int out_size, float scale, int sampling_ratio, int aligned,
                     float* __restrict__ gfeat) {
  extern __shared__ float tile[];  // [bins][C+1] grad tile
  __shared__ AxisW AX, AY;
  const int k = blockIdx.x;
  const int bins = out_size * out_size;
  const int ld = C + 1;
  const RoiGeom g = roi_geom(rois + (size_t)k * 5, out_size, scale, sampling_ratio, aligned, B);
  const float* gb = gout + (size_t)k * C * bins;
  for (int o = threadIdx.x; o < C * bins; o += blockDim.x) {
    const int c = o / bins, bin = o - c * bins;
    tile[bin * ld + c] = gb[o];
  }
  axis_weights(g.start_w, g.bin_w, g.grid_w, W, out_size, &AX);
  axis_weights(g.start_h, g.bin_h, g.grid_h, H, out_size, &AY);   // ends with a barrier
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

// ------------------------------------------------------------ unified out_size 7 path --
constexpr int SF = 5;      // side of the register-resident footprint (path A)
constexpr int GS = 16;     // most consecutive RoIs one workgroup takes
constexpr int MAXE = 104;  // per-axis extent (feature pixels) path B holds in LDS: every RoI on a 100x100 map (800 px tiles)

struct Roi7Lds {
  union {
    struct { float ax[GS][NB][SF]; float ay[GS][NB][SF]; } a;   // path A: per member, relative to the run's origin; ay / count
    struct { float wx[MAXE][8]; float wy[MAXE][8]; } b;         // path B: [pixel - origin][bin], one RoI at a time; wy / count
  } w;
  int ub[6];                  // path A: x0, y0, x1, y1 of the run's taps, batch index, ok
  int lo[2][NB], hi[2][NB];   // path B: band of every bin per axis (absolute pixel index; hi < lo: empty)
  int ext[4];                 // path B: ox, nx, oy, ny
};

__device__ __forceinline__ void small_tap(float v, int L, int& l, int& h, float& fl, bool& valid) {
  valid = !(v < -1.0f || v > (float)L);       // mirrors bilin(): v <= 0 -> 0; l >= L-1 -> l = h = L-1, frac 0
  if (v <= 0.f) v = 0.f;
  l = (int)v;
  if (l >= L - 1) { h = l = L - 1; v = (float)l; } else { h = l + 1; }
  fl = v - (float)l;
}

// Path A set-up for RoIs [k0, k0+n): exact union of the taps, then one thread per (member, axis, bin) writes that
// bin's <= 5 weights (no atomics).  S.ub[5] says whether the run qualifies.  Ends with a barrier.
__device__ void run_setup(const float* __restrict__ rois, int k0, int n, int B, int H, int W, float scale,
                          int sampling_ratio, int aligned, Roi7Lds& S, bool weights = true) {
  if (threadIdx.x == 0) { S.ub[0] = 1 << 30; S.ub[1] = 1 << 30; S.ub[2] = -1; S.ub[3] = -1; S.ub[4] = -1; S.ub[5] = 1; }
  __syncthreads();
  const int nt = n * 2 * NB;                       // <= 224 threads
  const int t = threadIdx.x;
  const int r = t / (2 * NB), a = (t / NB) & 1, p = t % NB;
  RoiGeom g;
  float start = 0.f, bin = 0.f;
  int grid = 0, L = 1;
  if (t < nt) {
    g = roi_geom(rois + (size_t)(k0 + r) * 5, NB, scale, sampling_ratio, aligned, B);
    start = a ? g.start_h : g.start_w; bin = a ? g.bin_h : g.bin_w;
    grid = a ? g.grid_h : g.grid_w; L = a ? H : W;
    if (grid > 8) S.ub[5] = 0;                     // such a bin alone spans more than the footprint
    for (int i = 0; i < grid && i < 8; ++i) {
      int l, h; float fl; bool valid;
      small_tap(start + p * bin + (i + .5f) * bin / (float)grid, L, l, h, fl, valid);
      if (!valid) continue;
      atomicMin(&S.ub[a], l);
      atomicMax(&S.ub[2 + a], h);
    }
    if (a == 0 && p == 0) {
      const int old = atomicCAS(&S.ub[4], -1, g.b);
      if (old != -1 && old != g.b) S.ub[5] = 0;    // the run spans two images
    }
  }
  __syncthreads();
  const int ox = S.ub[0], oy = S.ub[1];
  const bool ok = S.ub[5] && (S.ub[2] - ox < SF) && (S.ub[3] - oy < SF);
  __syncthreads();
  if (threadIdx.x == 0) S.ub[5] = ok ? 1 : 0;
  if (ok && weights && t < nt) {
    const int o = a ? oy : ox;
    float w[SF];
#pragma unroll
    for (int i = 0; i < SF; ++i) w[i] = 0.f;
    for (int i = 0; i < grid; ++i) {
      int l, h; float fl; bool valid;
      small_tap(start + p * bin + (i + .5f) * bin / (float)grid, L, l, h, fl, valid);
      if (!valid) continue;
#pragma unroll
      for (int j = 0; j < SF; ++j) {               // register array: static indices only
        if (j == l - o) w[j] += 1.f - fl;
        if (j == h - o) w[j] += fl;
      }
    }
    float* dst = a ? S.w.a.ay[r][p] : S.w.a.ax[r][p];
#pragma unroll
    for (int i = 0; i < SF; ++i) dst[i] = a ? w[i] / g.inv_count : w[i];
  }
  __syncthreads();
}

// Path B set-up for ONE RoI: bands, extent, dense per-pixel weight rows.  Every thread of the block calls it; ends
// with a barrier.  S.ext: nx or ny <= 0 -> the RoI samples nothing inside the map; > MAXE -> path C.
__device__ void roi_setup(const RoiGeom& g, int H, int W, Roi7Lds& S) {
  const int t = threadIdx.x, a = t / NB, p = t % NB;
  const float start = a ? g.start_h : g.start_w, bin = a ? g.bin_h : g.bin_w;
  const int grid = a ? g.grid_h : g.grid_w, L = a ? H : W;
  if (t < 2 * NB) {
    int lo = 1 << 30, hi = -1;
    for (int i = 0; i < grid; ++i) {
      int l, h; float fl; bool valid;
      small_tap(start + p * bin + (i + .5f) * bin / (float)grid, L, l, h, fl, valid);
      if (!valid) continue;
      lo = min(lo, l); hi = max(hi, h);
    }
    S.lo[a][p] = lo; S.hi[a][p] = hi;
  }
  __syncthreads();
  if (t < 2) {
    int o = 1 << 30, e = -1;
    for (int q = 0; q < NB; ++q)
      if (S.hi[t][q] >= S.lo[t][q]) { o = min(o, S.lo[t][q]); e = max(e, S.hi[t][q]); }
    S.ext[2 * t] = o; S.ext[2 * t + 1] = e >= o ? e - o + 1 : 0;
  }
  __syncthreads();
  const int ox = S.ext[0], nx = S.ext[1], oy = S.ext[2], ny = S.ext[3];
  if (nx <= 0 || ny <= 0 || nx > MAXE || ny > MAXE) return;      // block-uniform
  for (int i = t; i < (nx + ny) * 8; i += blockDim.x) {
    if (i < nx * 8) (&S.w.b.wx[0][0])[i] = 0.f; else (&S.w.b.wy[0][0])[i - nx * 8] = 0.f;
  }
  __syncthreads();
  if (t < 2 * NB) {                                 // column p of axis a belongs to this thread alone
    float (*w)[8] = a ? S.w.b.wy : S.w.b.wx;
    const int o = a ? oy : ox;
    for (int i = 0; i < grid; ++i) {
      int l, h; float fl; bool valid;
      small_tap(start + p * bin + (i + .5f) * bin / (float)grid, L, l, h, fl, valid);
      if (!valid) continue;
      w[l - o][p] += 1.f - fl;
      w[h - o][p] += fl;
    }
    if (a) for (int q = S.lo[1][p]; q <= S.hi[1][p]; ++q) w[q - o][p] = w[q - o][p] / g.inv_count;
  }
  __syncthreads();
}

// LDS tile [nc][49] (thread-major) <-> the contiguous nc*49 floats of one RoI's block, 16 bytes per lane.
__device__ __forceinline__ void tile_store(const float* __restrict__ tile, float* __restrict__ dst, int nf) {
  if (((nf & 3) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
    const float4* t4 = reinterpret_cast<const float4*>(tile);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = threadIdx.x; i < (nf >> 2); i += blockDim.x) d4[i] = t4[i];
  } else {
    for (int i = threadIdx.x; i < nf; i += blockDim.x) dst[i] = tile[i];
  }
}
// The same block as three bf16 planes (x = x0 + x1 + x2): what the FC stack's matrix kernel stages (csrc/gemm_split.hip), so
// neither the fp32 block nor a split pass over it ever touches HBM.  nf % 8 == 0, dst 16-byte aligned.
__device__ __forceinline__ void tile_store_planes(const float* __restrict__ tile, uint16_t* __restrict__ dst, long plane, int nf, int np) {
  for (int i = threadIdx.x; i < (nf >> 3); i += blockDim.x) {
    const float4 lo = *reinterpret_cast<const float4*>(tile + 8 * i), hi = *reinterpret_cast<const float4*>(tile + 8 * i + 4);
    if (np == 2) {                                   // two fp16 planes (pt_roi_align_fwd_planes_f16)
      uint4 h0, h1;
      split_pair_f16(lo.x, lo.y, h0.x, h1.x);
      split_pair_f16(lo.z, lo.w, h0.y, h1.y);
      split_pair_f16(hi.x, hi.y, h0.z, h1.z);
      split_pair_f16(hi.z, hi.w, h0.w, h1.w);
      *reinterpret_cast<uint4*>(dst + 8 * i) = h0;
      *reinterpret_cast<uint4*>(dst + plane + 8 * i) = h1;
      continue;
    }
    uint4 o0, o1, o2;
    split_pair(lo.x, lo.y, o0.x, o1.x, o2.x);
    split_pair(lo.z, lo.w, o0.y, o1.y, o2.y);
    split_pair(hi.x, hi.y, o0.z, o1.z, o2.z);
    split_pair(hi.z, hi.w, o0.w, o1.w, o2.w);
    *reinterpret_cast<uint4*>(dst + 8 * i) = o0;
    *reinterpret_cast<uint4*>(dst + plane + 8 * i) = o1;
    *reinterpret_cast<uint4*>(dst + 2 * plane + 8 * i) = o2;
  }
}
__device__ __forceinline__ void tile_load(float* __restrict__ tile, const float* __restrict__ src, int nf) {
  if (((nf & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
    float4* t4 = reinterpret_cast<float4*>(tile);
    const float4* s4 = reinterpret_cast<const float4*>(src);
    for (int i = threadIdx.x; i < (nf >> 2); i += blockDim.x) t4[i] = s4[i];
  } else {
    for (int i = threadIdx.x; i < nf; i += blockDim.x) tile[i] = src[i];
  }
}

// Path R: up to RN members of one run whose UNION spans <= RU pixels per axis.
constexpr int RN = 5;    // members held together
constexpr int RU = 48;   // largest per-axis extent (feature pixels) of the union
constexpr int RC = 64;   // channels per workgroup (one wavefront-wide slice)

struct RunRowsLds {
  float wx[RN][RU + 2][8];      // [member][pixel - union x0][bin]; zero outside the member (and for absent members)
  float wy[RN][RU + 2][8];      // [member][pixel - union y0][bin], divided by the sample count
  int lo[RN][2][NB], hi[RN][2][NB];
  int ext[RN][4];               // ox, nx, oy, ny per member (n <= 0: the member samples nothing inside the map)
  int u[4];                     // union of the members' footprints: x0, nx, y0, ny
  int b[RN];
  int ok;
};

// Set-up for members [k0, k0 + m): bands, extents, union, weights relative to the union's origin; R.ok (block-uniform after the
// final barrier) says whether the sub-run qualifies (union <= RU per axis, one image).  Mirrors roi_setup member by member.
__device__ void rows_setup(const float* __restrict__ rois, int k0, int m, int B, int H, int W, float scale, int sampling_ratio,
                           int aligned, RunRowsLds& R) {
  const int t = threadIdx.x;
  const int r = t / (2 * NB), a = (t / NB) & 1, p = t % NB;
  const bool mine = t < m * 2 * NB;
  RoiGeom g;
  float start = 0.f, bin = 0.f;
  int grid = 0, L = 1;
  if (mine) {
    g = roi_geom(rois + (size_t)(k0 + r) * 5, NB, scale, sampling_ratio, aligned, B);
    start = a ? g.start_h : g.start_w; bin = a ? g.bin_h : g.bin_w;
    grid = a ? g.grid_h : g.grid_w; L = a ? H : W;
    int lo = 1 << 30, hi = -1;
    for (int i = 0; i < grid; ++i) {
      int l, h; float fl; bool valid;
      small_tap(start + p * bin + (i + .5f) * bin / (float)grid, L, l, h, fl, valid);
      if (!valid) continue;
      lo = min(lo, l); hi = max(hi, h);
    }
    R.lo[r][a][p] = lo; R.hi[r][a][p] = hi;
    if (a == 0 && p == 0) R.b[r] = g.b;
  }
  for (int i = t; i < RN * (RU + 2) * 8; i += blockDim.x) { (&R.wx[0][0][0])[i] = 0.f; (&R.wy[0][0][0])[i] = 0.f; }
  __syncthreads();
  if (mine && p == 0) {
    int o = 1 << 30, e = -1;
    for (int q = 0; q < NB; ++q)
      if (R.hi[r][a][q] >= R.lo[r][a][q]) { o = min(o, R.lo[r][a][q]); e = max(e, R.hi[r][a][q]); }
    R.ext[r][2 * a] = o; R.ext[r][2 * a + 1] = e >= o ? e - o + 1 : 0;
  }
  __syncthreads();
  if (t == 0) {
    int x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1, ok = 1;
    for (int q = 0; q < m; ++q) {
      if (R.b[q] != R.b[0]) ok = 0;
      if (R.ext[q][1] > 0 && R.ext[q][3] > 0) {
        x0 = min(x0, R.ext[q][0]); x1 = max(x1, R.ext[q][0] + R.ext[q][1] - 1);
        y0 = min(y0, R.ext[q][2]); y1 = max(y1, R.ext[q][2] + R.ext[q][3] - 1);
      }
    }
    const int nx = x1 >= x0 ? x1 - x0 + 1 : 0, ny = y1 >= y0 ? y1 - y0 + 1 : 0;
    R.u[0] = x0; R.u[1] = nx; R.u[2] = y0; R.u[3] = ny;
    R.ok = ok && nx <= RU && ny <= RU;
  }
  __syncthreads();
  if (R.ok && mine && R.ext[r][1] > 0 && R.ext[r][3] > 0) {   // column p of axis a of member r belongs to this thread alone
    float (*w)[8] = a ? R.wy[r] : R.wx[r];
    const int o = a ? R.u[2] : R.u[0];
    for (int i = 0; i < grid; ++i) {
      int l, h; float fl; bool valid;
      small_tap(start + p * bin + (i + .5f) * bin / (float)grid, L, l, h, fl, valid);
      if (!valid) continue;
      w[l - o][p] += 1.f - fl;
      w[h - o][p] += fl;
    }
    if (a) for (int q = R.lo[r][1][p]; q <= R.hi[r][1][p]; ++q) w[q - o][p] = w[q - o][p] / g.inv_count;
  }
  __syncthreads();
}

// Path R for one 64-channel slice: tiles[m][64][49] in LDS, wavefront w takes the union's rows w, w + 4, ...; a row's pixels go
// two at a time with every member's weights loaded unconditionally (zeros outside a member): no branches in the pixel loop.
__device__ void rows_path(const float* __restrict__ gout, int k0, int m, int C, int H, int W, int c0, float* __restrict__ tiles,
                          const RunRowsLds& R, float* __restrict__ gfeat) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int r = 0; r < m; ++r) tile_load(tiles + r * RC * 49, gout + ((size_t)(k0 + r) * C + c0) * 49, RC * 49);
  __syncthreads();
  const int x0 = R.u[0], nx = R.u[1], y0 = R.u[2], ny = R.u[3];
  float* fb = gfeat + ((size_t)R.b[0] * H * W + x0) * C + c0 + lane;
  for (int py = wv; py < ny; py += 4) {
    float s7[RN][NB];
#pragma unroll
    for (int r = 0; r < RN; ++r) {
#pragma unroll
      for (int pw = 0; pw < NB; ++pw) s7[r][pw] = 0.f;
      if (r < m) {                                                                     // wave-uniform
        const float* my = tiles + (r * RC + lane) * 49;
        const float4 ya = *reinterpret_cast<const float4*>(&R.wy[r][py][0]);
        const float4 yb = *reinterpret_cast<const float4*>(&R.wy[r][py][4]);
        const float wy[NB] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z};
#pragma unroll
        for (int ph = 0; ph < NB; ++ph) {
          if (__builtin_amdgcn_readfirstlane(__float_as_uint(wy[ph])) != 0u) {       // a row meets one or two bins of a large RoI
#pragma unroll
            for (int pw = 0; pw < NB; ++pw) s7[r][pw] = fmaf(wy[ph], my[ph * NB + pw], s7[r][pw]);
          }
        }
      }
    }
    float* row = fb + (size_t)(y0 + py) * W * C;
    for (int px = 0; px < nx; px += 2) {                                               // wx has two zero rows of padding
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int r = 0; r < RN; ++r) {
        const float4 a0 = *reinterpret_cast<const float4*>(&R.wx[r][px][0]);
        const float4 b0 = *reinterpret_cast<const float4*>(&R.wx[r][px][4]);
        const float4 a1 = *reinterpret_cast<const float4*>(&R.wx[r][px + 1][0]);
        const float4 b1 = *reinterpret_cast<const float4*>(&R.wx[r][px + 1][4]);
        v0 += a0.x * s7[r][0] + a0.y * s7[r][1] + a0.z * s7[r][2] + a0.w * s7[r][3] + b0.x * s7[r][4] + b0.y * s7[r][5] + b0.z * s7[r][6];
        v1 += a1.x * s7[r][0] + a1.y * s7[r][1] + a1.z * s7[r][2] + a1.w * s7[r][3] + b1.x * s7[r][4] + b1.y * s7[r][5] + b1.z * s7[r][6];
      }
      if (v0 != 0.f) atomicAdd(&row[(size_t)px * C], v0);
      if (v1 != 0.f && px + 1 < nx) atomicAdd(&row[(size_t)(px + 1) * C], v1);
    }
  }
  __syncthreads();                                       // the tiles are reloaded for the next slice / sub-run
}

__global__ void __launch_bounds__(256)
    roi_align7_fwd(const float* __restrict__ feat, const float* __restrict__ rois, int B, int C, int H, int W, int K,
                   int gs, float scale, int sampling_ratio, int aligned, float* __restrict__ out, int tile_bytes,
                   uint16_t* __restrict__ out_planes, long plane, int np) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);                       // [<=256][49]
  Roi7Lds& S = *reinterpret_cast<Roi7Lds*>(smem + tile_bytes);
  const int k0 = blockIdx.x * gs, n = min(gs, K - k0);
  if (out_planes && blockIdx.x == 0) {                                 // row K of the planes: the zero row the matrix kernels read
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int i = threadIdx.x; i < (C * 49) >> 3; i += blockDim.x)
      for (int p = 0; p < np; ++p) *reinterpret_cast<uint4*>(out_planes + p * plane + (size_t)K * C * 49 + 8 * i) = z;
  }
  run_setup(rois, k0, n, B, H, W, scale, sampling_ratio, aligned, S);
  const bool path_a = S.ub[5] != 0;
  const int ux = S.ub[0], uy = S.ub[1], ub = S.ub[4];
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int nc = min(256, C - c0), c = c0 + threadIdx.x;
    const bool act = (int)threadIdx.x < nc;
    float* my = tile + threadIdx.x * 49;
    if (path_a) {
      float f[SF][SF];
#pragma unroll
      for (int y = 0; y < SF; ++y)
#pragma unroll
        for (int x = 0; x < SF; ++x) {
          const int yy = min(uy + y, H - 1), xx = min(ux + x, W - 1);   // rows / columns beyond the union carry zero weight
          f[y][x] = act ? feat[(((size_t)ub * H + yy) * W + xx) * C + c] : 0.f;
        }
      for (int r = 0; r < n; ++r) {
        if (act) {
          float T[SF][NB];
#pragma unroll
          for (int y = 0; y < SF; ++y)
#pragma unroll
            for (int pw = 0; pw < NB; ++pw) {
              float t = 0.f;
#pragma unroll
              for (int x = 0; x < SF; ++x) t = fmaf(S.w.a.ax[r][pw][x], f[y][x], t);
              T[y][pw] = t;
            }
#pragma unroll
          for (int ph = 0; ph < NB; ++ph)
#pragma unroll
            for (int pw = 0; pw < NB; ++pw) {
              float v = 0.f;
#pragma unroll
              for (int y = 0; y < SF; ++y) v = fmaf(S.w.a.ay[r][ph][y], T[y][pw], v);
              my[ph * NB + pw] = v;
            }
        }
        __syncthreads();
        if (out_planes) tile_store_planes(tile, out_planes + ((size_t)(k0 + r) * C + c0) * 49, plane, nc * 49, np);
        else tile_store(tile, out + ((size_t)(k0 + r) * C + c0) * 49, nc * 49);
        __syncthreads();
      }
    } else {
      for (int r = 0; r < n; ++r) {
        const RoiGeom g = roi_geom(rois + (size_t)(k0 + r) * 5, NB, scale, sampling_ratio, aligned, B);
        roi_setup(g, H, W, S);
        const int ox = S.ext[0], nx = S.ext[1], oy = S.ext[2], ny = S.ext[3];
        if (act) {
          const float* fb = feat + (size_t)g.b * H * W * C + c;
          if (nx <= 0 || ny <= 0) {
#pragma unroll
            for (int i = 0; i < 49; ++i) my[i] = 0.f;
          } else if (nx <= MAXE && ny <= MAXE) {                        // path B: separable sums
            // 4 footprint rows x 8 pixels = 32 independent 1-KiB row reads in flight per wave (the map is L2 / MALL
            // resident: this path is bound by load latency, not by bytes); the 7 column weights of a pixel are read
            // once (two broadcast ds_read_b128) and serve the 4 rows; a row then updates only the row-bins whose band
            // holds it (wave-uniform test).
            float acc[NB][NB];
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
              for (int b = 0; b < NB; ++b) acc[a][b] = 0.f;
            int ylo[NB], yhi[NB];
#pragma unroll
            for (int a = 0; a < NB; ++a) { ylo[a] = S.lo[1][a] - oy; yhi[a] = S.hi[1][a] - oy; }
            for (int py0 = 0; py0 < ny; py0 += 4) {
              float t[4][NB];
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int b = 0; b < NB; ++b) t[r4][b] = 0.f;
              const float* row0 = fb + ((size_t)(oy + py0) * W + ox) * C;
              for (int px0 = 0; px0 < nx; px0 += 8) {
                float v[4][8];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                  for (int j = 0; j < 8; ++j) {
                    const bool in = (py0 + r4 < ny) && (px0 + j < nx);                        // wave-uniform
                    v[r4][j] = in ? row0[((size_t)r4 * W + px0 + j) * C] : 0.f;
                  }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  if (px0 + j < nx) {
                    const float4 wa = *reinterpret_cast<const float4*>(&S.w.b.wx[px0 + j][0]);
                    const float4 wb = *reinterpret_cast<const float4*>(&S.w.b.wx[px0 + j][4]);
                    const float w7[NB] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z};
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                      for (int b = 0; b < NB; ++b) t[r4][b] = fmaf(w7[b], v[r4][j], t[r4][b]);
                  }
                }
              }
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                const int py = py0 + r4;
                if (py < ny) {
#pragma unroll
                  for (int a = 0; a < NB; ++a) {
                    if (py >= ylo[a] && py <= yhi[a]) {
                      const float wy = S.w.b.wy[py][a];
#pragma unroll
                      for (int b = 0; b < NB; ++b) acc[a][b] = fmaf(wy, t[r4][b], acc[a][b]);
                    }
                  }
                }
              }
            }
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
              for (int b = 0; b < NB; ++b) my[a * NB + b] = acc[a][b];
          } else {                                                      // path C: direct taps
            const float* fc = fb;
            for (int ph = 0; ph < NB; ++ph)
              for (int pw = 0; pw < NB; ++pw) {
                float acc = 0.f;
                for (int iy = 0; iy < g.grid_h; ++iy) {
                  const float y = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.grid_h;
                  for (int ix = 0; ix < g.grid_w; ++ix) {
                    const float x = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.grid_w;
                    const Bilin q = bilin(y, x, H, W);
                    if (q.valid)
                      acc += q.w1 * fc[((size_t)q.y0 * W + q.x0) * C] + q.w2 * fc[((size_t)q.y0 * W + q.x1) * C] +
                             q.w3 * fc[((size_t)q.y1 * W + q.x0) * C] + q.w4 * fc[((size_t)q.y1 * W + q.x1) * C];
                  }
                }
                my[ph * NB + pw] = acc / g.inv_count;
              }
          }
        }
        __syncthreads();
        if (out_planes) tile_store_planes(tile, out_planes + ((size_t)(k0 + r) * C + c0) * 49, plane, nc * 49, np);
        else tile_store(tile, out + ((size_t)(k0 + r) * C + c0) * 49, nc * 49);
        __syncthreads();
      }
    }
  }
}

// Backward of ONE member on path B (separable weights, one atomic per footprint pixel and channel) or C (direct taps).  The
// caller's workgroups share a path-B member by footprint rows (rows phase, phase + step, ...: a 60 x 60 member is 3 600 serial
// atomics per thread otherwise - the launch's tail); path C belongs to phase 0 alone.
__device__ void member_bc(const float* __restrict__ gout, const float* __restrict__ rois, int k, int B, int C, int H, int W,
                          float scale, int sampling_ratio, int aligned, float* __restrict__ tile, Roi7Lds& S,
                          float* __restrict__ gfeat, int phase, int step) {
  const RoiGeom g = roi_geom(rois + (size_t)k * 5, NB, scale, sampling_ratio, aligned, B);
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int nc = min(256, C - c0), c = c0 + threadIdx.x;
    const bool act = (int)threadIdx.x < nc;
    const float* my = tile + threadIdx.x * 49;
    tile_load(tile, gout + ((size_t)k * C + c0) * 49, nc * 49);
    roi_setup(g, H, W, S);                                          // its barriers also publish the tile
    const int ox = S.ext[0], nx = S.ext[1], oy = S.ext[2], ny = S.ext[3];
    if (act && nx > 0 && ny > 0) {
      float* fb = gfeat + (size_t)g.b * H * W * C + c;
      if (nx <= MAXE && ny <= MAXE) {                               // path B
        float gv[49];
#pragma unroll
        for (int i = 0; i < 49; ++i) gv[i] = my[i];
        for (int py = phase; py < ny; py += step) {
          const float4 ya = *reinterpret_cast<const float4*>(&S.w.b.wy[py][0]);
          const float4 yb = *reinterpret_cast<const float4*>(&S.w.b.wy[py][4]);
          const float wy[NB] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z};
          float s7[NB];
#pragma unroll
          for (int pw = 0; pw < NB; ++pw) {
            float t = 0.f;
#pragma unroll
            for (int ph = 0; ph < NB; ++ph) t = fmaf(wy[ph], gv[ph * NB + pw], t);
            s7[pw] = t;
          }
          float* row = fb + ((size_t)(oy + py) * W + ox) * C;
          for (int px = 0; px < nx; px += 2) {                      // two pixels per trip: the weight reads overlap
            const int p1 = min(px + 1, MAXE - 1);
            const float4 wa = *reinterpret_cast<const float4*>(&S.w.b.wx[px][0]);
            const float4 wb = *reinterpret_cast<const float4*>(&S.w.b.wx[px][4]);
            const float4 wc = *reinterpret_cast<const float4*>(&S.w.b.wx[p1][0]);
            const float4 wd = *reinterpret_cast<const float4*>(&S.w.b.wx[p1][4]);
            const float v0 = wa.x * s7[0] + wa.y * s7[1] + wa.z * s7[2] + wa.w * s7[3] + wb.x * s7[4] + wb.y * s7[5] +
                             wb.z * s7[6];
            const float v1 = wc.x * s7[0] + wc.y * s7[1] + wc.z * s7[2] + wc.w * s7[3] + wd.x * s7[4] + wd.y * s7[5] +
                             wd.z * s7[6];
            if (v0 != 0.f) atomicAdd(&row[(size_t)px * C], v0);
            if (v1 != 0.f && px + 1 < nx) atomicAdd(&row[(size_t)(px + 1) * C], v1);
          }
        }
      } else if (phase == 0) {                                      // path C: direct taps
        for (int ph = 0; ph < NB; ++ph)
          for (int pw = 0; pw < NB; ++pw) {
            const float gvv = my[ph * NB + pw] / g.inv_count;
            for (int iy = 0; iy < g.grid_h; ++iy) {
              const float y = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.grid_h;
              for (int ix = 0; ix < g.grid_w; ++ix) {
                const float x = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.grid_w;
                const Bilin q = bilin(y, x, H, W);
                if (q.valid) {
                  atomicAdd(&fb[((size_t)q.y0 * W + q.x0) * C], gvv * q.w1);
                  atomicAdd(&fb[((size_t)q.y0 * W + q.x1) * C], gvv * q.w2);
                  atomicAdd(&fb[((size_t)q.y1 * W + q.x0) * C], gvv * q.w3);
                  atomicAdd(&fb[((size_t)q.y1 * W + q.x1) * C], gvv * q.w4);
                }
              }
            }
          }
      }
    }
    __syncthreads();                                                // the tile is reloaded for the next slice / RoI
  }
}

__global__ void __launch_bounds__(256)
    roi_align7_bwd(const float* __restrict__ gout, const float* __restrict__ rois, int B, int C, int H, int W, int K,
                   int gs, float scale, int sampling_ratio, int aligned, float* __restrict__ gfeat, int tile_bytes) {
  // grid (runs, max(gs, C / 64)): a run whose taps fit the register footprint (path A) is reduced by its workgroup y == 0 alone
  // (one atomic pass for the whole run).  Any other run is cut into sub-runs of up to RN members: where the members qualify
  // (path R) workgroup y takes the 64-channel slices y, y + gridDim.y, ... and adds the members in registers before the atomics
  // of the union's pixels; otherwise every workgroup of the run takes its share of each member's footprint rows on path B, so
  // large RoIs - whose cost is their footprint's worth of atomics - are spread over workgroups instead of queueing in one.
  extern __shared__ __align__(16) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  Roi7Lds& S = *reinterpret_cast<Roi7Lds*>(smem + tile_bytes);
  const int k0 = blockIdx.x * gs, n = min(gs, K - k0);
  const int y = blockIdx.y, gy = gridDim.y;
  run_setup(rois, k0, n, B, H, W, scale, sampling_ratio, aligned, S, y == 0);
  const bool path_a = S.ub[5] != 0;
  if (!path_a) {
    // rows_setup gives every workgroup of the run the same answer (it depends on the RoIs alone)
    RunRowsLds& R = *reinterpret_cast<RunRowsLds*>(smem + RN * RC * 49 * 4);
    for (int r0 = 0; r0 < n; r0 += RN) {
      const int m = min(RN, n - r0);
      bool rows = m >= 2 && (C % RC) == 0;
      if (rows) {
        __syncthreads();                                            // S / the tile are dead: R aliases them
        rows_setup(rois, k0 + r0, m, B, H, W, scale, sampling_ratio, aligned, R);
        rows = R.ok != 0;
      }
      if (rows) {
        if (R.u[1] > 0 && R.u[3] > 0)
          for (int sl = y; sl < C / RC; sl += gy) rows_path(gout, k0 + r0, m, C, H, W, sl * RC, tile, R, gfeat);
      } else {
        __syncthreads();
        for (int r = r0; r < r0 + m; ++r)
          member_bc(gout, rois, k0 + r, B, C, H, W, scale, sampling_ratio, aligned, tile, S, gfeat, y, gy);
      }
    }
    return;
  }
  if (y != 0) return;
  const int ux = S.ub[0], uy = S.ub[1], ub = S.ub[4];
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int nc = min(256, C - c0), c = c0 + threadIdx.x;
    const bool act = (int)threadIdx.x < nc;
    const float* my = tile + threadIdx.x * 49;
    float acc[SF][SF];
#pragma unroll
    for (int y = 0; y < SF; ++y)
#pragma unroll
      for (int x = 0; x < SF; ++x) acc[y][x] = 0.f;
    for (int r = 0; r < n; ++r) {
      tile_load(tile, gout + ((size_t)(k0 + r) * C + c0) * 49, nc * 49);
      __syncthreads();
      if (act) {
        float Sy[SF][NB];
#pragma unroll
        for (int y = 0; y < SF; ++y)
#pragma unroll
          for (int pw = 0; pw < NB; ++pw) Sy[y][pw] = 0.f;
#pragma unroll
        for (int ph = 0; ph < NB; ++ph)
#pragma unroll
          for (int pw = 0; pw < NB; ++pw) {
            const float gv = my[ph * NB + pw];
#pragma unroll
            for (int y = 0; y < SF; ++y) Sy[y][pw] = fmaf(S.w.a.ay[r][ph][y], gv, Sy[y][pw]);
          }
#pragma unroll
        for (int y = 0; y < SF; ++y)
#pragma unroll
          for (int x = 0; x < SF; ++x) {
            float v = acc[y][x];
#pragma unroll
            for (int pw = 0; pw < NB; ++pw) v = fmaf(S.w.a.ax[r][pw][x], Sy[y][pw], v);
            acc[y][x] = v;
          }
      }
      __syncthreads();
    }
    if (act) {
#pragma unroll
      for (int y = 0; y < SF; ++y)
#pragma unroll
        for (int x = 0; x < SF; ++x) {
          const float v = acc[y][x];
          if (v != 0.f && uy + y < H && ux + x < W)
            atomicAdd(&gfeat[(((size_t)ub * H + uy + y) * W + ux + x) * C + c], v);
        }
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
  if (channels_last && out_size != NB) {
    PT_REQUIRE((size_t)(C + 1) * out_size * out_size * 4 + 2 * sizeof(pt::AxisW) <= 150 * 1024, PT_ELIMIT,
               "%s: C=%d too large for the LDS tile", fn, C);
    PT_REQUIRE(H <= pt::MAXR && W <= pt::MAXR, PT_ELIMIT, "%s: out_size != 7 supports maps up to %dx%d", fn,
               pt::MAXR, pt::MAXR);
  }
  return PT_OK;
}

// The out_size-7 kernels always ask for the same dynamic LDS (a [256][49] tile + Roi7Lds), so the attribute is
// set once per process through a thread-safe function-local static.
static constexpr int ROI7_TILE_BYTES = 256 * 49 * 4;
static constexpr int ROI7_LDS = ROI7_TILE_BYTES + (int)((sizeof(pt::Roi7Lds) + 15) / 16 * 16);
// backward: the larger of that and path R's [RN][64][49] tiles + RunRowsLds (both under 80 KiB: two workgroups per CU)
static constexpr int ROI7_ROWS_LDS = pt::RN * pt::RC * 49 * 4 + (int)((sizeof(pt::RunRowsLds) + 15) / 16 * 16);
static constexpr int ROI7_BWD_LDS = ROI7_ROWS_LDS > ROI7_LDS ? ROI7_ROWS_LDS : ROI7_LDS;
static_assert(ROI7_BWD_LDS <= 80 * 1024, "two backward workgroups per CU");
static hipError_t roi7_attr() {
  static const hipError_t rc = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align7_fwd),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, ROI7_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align7_bwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                               ROI7_BWD_LDS);
  }();
  return rc;
}
static hipError_t roi_generic_attr() {
  static const hipError_t rc = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_cl),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_bwd_cl), hipFuncAttributeMaxDynamicSharedMemorySize,
                               100 * 1024);
  }();
  return rc;
}

// Run length = a divisor of the bag size (runs never straddle two bags), as long as possible while the launch keeps
// at least `min_wgs` workgroups.  The forward shares only L2-resident loads inside a run, so it prefers many
// workgroups (2048 = 4 per CU at 2 resident); the backward saves one atomic pass per run, so it prefers long runs.
static int run_length(int group, int K, int min_wgs) {
  if (group < 1) group = 1;
  int gs = 1;
  for (int d = 1; d <= GS; ++d)
    if (group % d == 0 && (d == 1 || K / d >= min_wgs)) gs = d;
  return gs;
}

extern "C" int pt_roi_align_fwd(const float* feat, const float* rois, int B, int C, int H, int W, int K, int out_size,
                                float spatial_scale, int sampling_ratio, int aligned, int channels_last, int group,
                                float* out, void* stream) {
  if (K == 0) return PT_OK;
  int rc = roi_check("pt_roi_align_fwd", feat, rois, out, B, C, H, W, K, out_size, channels_last);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  if (channels_last && out_size == NB) {
    hipError_t e = roi7_attr();
    if (e != hipSuccess) { set_error("pt_roi_align_fwd: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    const int gs = run_length(group, K, 2048);
    hipLaunchKernelGGL(roi_align7_fwd, dim3(cdiv(K, gs)), dim3(256), ROI7_LDS, s, feat, rois, B, C, H, W, K, gs,
                       spatial_scale, sampling_ratio, aligned, out, ROI7_TILE_BYTES, (uint16_t*)nullptr, 0L, 3);
  } else if (channels_last) {
    const size_t lds = (size_t)(C + 1) * out_size * out_size * sizeof(float);
    hipError_t e = roi_generic_attr();
    if (e != hipSuccess) { set_error("pt_roi_align_fwd: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    PT_REQUIRE(lds <= 100 * 1024, PT_ELIMIT, "pt_roi_align_fwd: C=%d too large for the LDS tile", C);
    hipLaunchKernelGGL(roi_align_fwd_cl, dim3(K), dim3(256), lds, s, feat, rois, B, C, H, W, out_size, spatial_scale,
                       sampling_ratio, aligned, out);
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

static int roi_fwd_planes(const char* fn, int np, const float* feat, const float* rois, int B, int C, int H, int W, int K, float spatial_scale,
                          int sampling_ratio, int aligned, int group, uint16_t* planes, int64_t plane_stride, void* stream) {
  PT_REQUIRE(planes && K >= 0 && C > 0 && (C * 49) % 8 == 0 && plane_stride >= (int64_t)(K + 1) * C * 49 && (plane_stride & 7) == 0 &&
                 (((uintptr_t)planes) & 15) == 0,
             PT_EINVAL, "pt_roi_align_fwd_planes: planes of (K + 1) * C * 49 elements each, 16-byte aligned, C * 49 a multiple of 8");
  if (K == 0) {
    for (int p = 0; p < np; ++p)
      if (hipMemsetAsync(planes + p * plane_stride, 0, (size_t)C * 49 * 2, as_stream(stream)) != hipSuccess) return PT_EINVAL;
    return PT_OK;
  }
  int rc = roi_check(fn, feat, rois, planes, B, C, H, W, K, NB, 1);
  if (rc) return rc;
  hipError_t e = roi7_attr();
  if (e != hipSuccess) { set_error("pt_roi_align_fwd_planes: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  const int gs = run_length(group, K, 2048);
  hipLaunchKernelGGL(roi_align7_fwd, dim3(cdiv(K, gs)), dim3(256), ROI7_LDS, as_stream(stream), feat, rois, B, C, H, W, K, gs, spatial_scale,
                     sampling_ratio, aligned, (float*)nullptr, ROI7_TILE_BYTES, planes, (long)plane_stride, np);
  PT_LAUNCH_CHECK(fn);
  // the scaled fp16 format (ABI 6): a plane stride with room for the tail gets 1 / scale = 1 written behind the zero row
  if (np == 2 && plane_stride >= (int64_t)(K + 1) * C * 49 + 8) {
    if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(planes + (int64_t)(K + 1) * C * 49), 0x3f800000, 1, as_stream(stream)) != hipSuccess) {
      set_error("%s: tail write failed", fn);
      return PT_EINVAL;
    }
  }
  return PT_OK;
}

extern "C" int pt_roi_align_fwd_planes(const float* feat, const float* rois, int B, int C, int H, int W, int K, float spatial_scale,
                                       int sampling_ratio, int aligned, int group, uint16_t* planes, int64_t plane_stride,
                                       void* stream) {
  return roi_fwd_planes("pt_roi_align_fwd_planes", 3, feat, rois, B, C, H, W, K, spatial_scale, sampling_ratio, aligned, group, planes,
                        plane_stride, stream);
}

extern "C" int pt_roi_align_fwd_planes_f16(const float* feat, const float* rois, int B, int C, int H, int W, int K, float spatial_scale,
                                           int sampling_ratio, int aligned, int group, uint16_t* planes, int64_t plane_stride,
                                           void* stream) {
  return roi_fwd_planes("pt_roi_align_fwd_planes_f16", 2, feat, rois, B, C, H, W, K, spatial_scale, sampling_ratio, aligned, group, planes,
                        plane_stride, stream);
}

extern "C" int pt_roi_align_bwd(const float* grad_out, const float* rois, int B, int C, int H, int W, int K,
                                int out_size, float spatial_scale, int sampling_ratio, int aligned,
                                int channels_last, int group, float* grad_feat, void* stream) {
  if (K == 0) return PT_OK;
  int rc = roi_check("pt_roi_align_bwd", grad_out, rois, grad_feat, B, C, H, W, K, out_size, channels_last);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  if (channels_last && out_size == NB) {
    hipError_t e = roi7_attr();
    if (e != hipSuccess) { set_error("pt_roi_align_bwd: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    const int gs = run_length(group, K, 512);
    const int slices = (C % pt::RC) == 0 ? C / pt::RC : 1;
    // (dealing the runs of one bag apart in dispatch order - so that their atomics do not queue on the same pixels - was tried:
    // the large-bag launch gained 12 %, the launches of tiny bags lost 20 %: kept adjacent)
    hipLaunchKernelGGL(roi_align7_bwd, dim3(cdiv(K, gs), gs > slices ? gs : slices), dim3(256), ROI7_BWD_LDS, s, grad_out, rois, B, C,
                       H, W, K, gs, spatial_scale, sampling_ratio, aligned, grad_feat, ROI7_TILE_BYTES);
  } else if (channels_last) {
    const size_t lds = (size_t)(C + 1) * out_size * out_size * sizeof(float);
    hipError_t e = roi_generic_attr();
    if (e != hipSuccess) { set_error("pt_roi_align_bwd: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    PT_REQUIRE(lds <= 100 * 1024, PT_ELIMIT, "pt_roi_align_bwd: C=%d too large for the LDS tile", C);
    hipLaunchKernelGGL(roi_align_bwd_cl, dim3(K), dim3(256), lds, s, grad_out, rois, B, C, H, W, K, 1,
                       out_size, spatial_scale, sampling_ratio, aligned, grad_feat);
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
