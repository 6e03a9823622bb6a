// Oriented-box kernels for the OBB variant of the path (OBB_TOD/mmrotate, config 5):
//   * differentiable rotated IoU  (replaces mmcv.ops.diff_iou_rotated_2d, call sites
//     OBB_TOD/mmrotate/models/losses/rotated_iou_loss.py:47,90)
//   * RoIAlignRotated fwd/bwd      (replaces mmcv.ops.RoIAlignRotated, call site
//     OBB_TOD/mmrotate/models/roi_heads/roi_extractors/rotate_single_level_roi_extractor.py:126)
// The IoU gradient is obtained by running the same convex-clip code on forward-mode dual numbers
// (value + 5 partials w.r.t. the predicted box), one thread per box pair: no sort kernel, no
// intermediate vertex tensors (mmcv materialises [B,N,24,2] vertices and a sort index).
#include "pt_common.h"

namespace pt {

// ------------------------------------------------------------------ dual numbers --
template <int ND>
struct Dual {
  float v;
  float d[ND];
  __device__ Dual() {}
  __device__ Dual(float x) : v(x) {
#pragma unroll
    for (int i = 0; i < ND; ++i) d[i] = 0.f;
  }
};
template <int ND>
__device__ __forceinline__ Dual<ND> operator+(const Dual<ND>& a, const Dual<ND>& b) {
  Dual<ND> r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
template <int ND>
__device__ __forceinline__ Dual<ND> operator-(const Dual<ND>& a, const Dual<ND>& b) {
  Dual<ND> r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
template <int ND>
__device__ __forceinline__ Dual<ND> operator*(const Dual<ND>& a, const Dual<ND>& b) {
  Dual<ND> r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
template <int ND>
__device__ __forceinline__ Dual<ND> operator/(const Dual<ND>& a, const Dual<ND>& b) {
  Dual<ND> r; const float inv = 1.f / b.v; r.v = a.v * inv;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
__device__ __forceinline__ float val(float x) { return x; }
template <int ND>
__device__ __forceinline__ float val(const Dual<ND>& x) { return x.v; }

template <typename T>
struct Pt {
  T x, y;
};

// Intersection area of the convex quad `p1` (type T: float or dual) with the float quad `p2`
// (counter-clockwise in image axes), Sutherland-Hodgman + shoelace.
template <typename T>
__device__ T clip_area(const Pt<T>* p1, const Pt<float>* p2) {
  Pt<T> A[10], Bf[10];
  int n = 4;
  for (int i = 0; i < 4; ++i) A[i] = p1[i];
  Pt<T>* cur = A;
  Pt<T>* nxt = Bf;
  for (int e = 0; e < 4 && n > 0; ++e) {
    const Pt<float> a = p2[e], b = p2[(e + 1) & 3];
    const float ex = b.x - a.x, ey = b.y - a.y;
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const Pt<T> p = cur[i], q = cur[(i + 1 == n) ? 0 : i + 1];
      const T sp = (p.y - T(a.y)) * T(ex) - (p.x - T(a.x)) * T(ey);
      const T sq = (q.y - T(a.y)) * T(ex) - (q.x - T(a.x)) * T(ey);
      if (val(sp) >= 0.f) nxt[m++] = p;
      if ((val(sp) > 0.f && val(sq) < 0.f) || (val(sp) < 0.f && val(sq) > 0.f)) {
        const T t = sp / (sp - sq);
        nxt[m].x = p.x + t * (q.x - p.x);
        nxt[m].y = p.y + t * (q.y - p.y);
        ++m;
      }
    }
    n = m;
    Pt<T>* tmp = cur; cur = nxt; nxt = tmp;
  }
  T area(0.f);
  for (int i = 0; i < n; ++i) {
    const Pt<T> p = cur[i], q = cur[(i + 1 == n) ? 0 : i + 1];
    area = area + (p.x * q.y - q.x * p.y);
  }
  return area * T(0.5f);
}

template <typename T>
__device__ __forceinline__ void corners(const T cx, const T cy, const T w, const T h, const T c, const T s, Pt<T>* p) {
  const T hw = w * T(0.5f), hh = h * T(0.5f);
  const float sx[4] = {-1.f, 1.f, 1.f, -1.f}, sy[4] = {-1.f, -1.f, 1.f, 1.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const T dx = hw * T(sx[i]), dy = hh * T(sy[i]);
    p[i].x = cx + dx * c - dy * s;
    p[i].y = cy + dx * s + dy * c;
  }
}

__global__ void diff_iou_rotated_kernel(const float* __restrict__ b1, const float* __restrict__ b2, int N,
                                        const float* __restrict__ gout, float* __restrict__ iou,
                                        float* __restrict__ grad1) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* a = b1 + (size_t)n * 5;
  const float* b = b2 + (size_t)n * 5;
  // box 2 relative to box 1's centre (fp32 precision), float corners
  Pt<float> q[4];
  corners<float>(b[0] - a[0], b[1] - a[1], b[2], b[3], cosf(b[4]), sinf(b[4]), q);
  const float a2 = b[2] * b[3];
  if (!grad1) {
    Pt<float> p[4];
    corners<float>(0.f, 0.f, a[2], a[3], cosf(a[4]), sinf(a[4]), p);
    const float inter = fabsf(clip_area<float>(p, q));
    const float a1 = a[2] * a[3];
    iou[n] = (a1 < 1e-14f || a2 < 1e-14f) ? 0.f : inter / (a1 + a2 - inter);
    return;
  }
  typedef Dual<5> D;
  D cx(0.f), cy(0.f), w(a[2]), h(a[3]), c(cosf(a[4])), s(sinf(a[4]));
  cx.d[0] = 1.f; cy.d[1] = 1.f; w.d[2] = 1.f; h.d[3] = 1.f;
  c.d[4] = -sinf(a[4]); s.d[4] = cosf(a[4]);
  Pt<D> p[4];
  corners<D>(cx, cy, w, h, c, s, p);
  D inter = clip_area<D>(p, q);
  if (inter.v < 0.f) inter = D(0.f) - inter;
  const D a1 = w * h;
  const D io = inter / (a1 + D(a2) - inter);
  const float g = gout[n];
  const bool dead = (a1.v < 1e-14f || a2 < 1e-14f);
#pragma unroll
  for (int k = 0; k < 5; ++k) grad1[(size_t)n * 5 + k] = dead ? 0.f : g * io.d[k];
}

// ------------------------------------------------------------ RoIAlignRotated ----
struct RRoi {
  int b;
  float cw, ch, start_w, start_h, bin_w, bin_h, cosv, sinv, count;
  int grid_w, grid_h;
};

__device__ __forceinline__ RRoi rroi_geom(const float* __restrict__ r, int out_size, float scale, int sample_num,
                                          int aligned, int clockwise, int B) {
  RRoi g;
  g.b = min(max((int)r[0], 0), B - 1);
  const float off = aligned ? 0.5f : 0.f;
  g.cw = r[1] * scale - off;
  g.ch = r[2] * scale - off;
  float rw = r[3] * scale, rh = r[4] * scale;
  float theta = r[5];
  if (clockwise) theta = -theta;
  if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  g.bin_h = rh / (float)out_size;
  g.bin_w = rw / (float)out_size;
  g.grid_h = sample_num > 0 ? sample_num : (int)ceilf(rh / (float)out_size);
  g.grid_w = sample_num > 0 ? sample_num : (int)ceilf(rw / (float)out_size);
  g.start_h = -rh / 2.f;
  g.start_w = -rw / 2.f;
  g.cosv = cosf(theta);
  g.sinv = sinf(theta);
  g.count = fmaxf((float)(g.grid_h * g.grid_w), 1.f);
  return g;
}

struct Tap4 {
  int y0, y1, x0, x1;
  float w1, w2, w3, w4;
  bool valid;
};
__device__ __forceinline__ Tap4 tap4(float y, float x, int H, int W) {
  Tap4 r;
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

// Generic over the two layouts through element strides (sc, sy, sx) and the thread->element map:
// CL = 1: consecutive threads are consecutive channels (coalesced [B,H,W,C] reads / atomics).
template <bool BWD, bool CL>
__global__ void __launch_bounds__(256)
    roi_align_rotated_kernel(const float* __restrict__ src, const float* __restrict__ rois, int B, long total, int C,
                             int H, int W, int out_size, float scale, int sample_num, int aligned, int clockwise,
                             float* __restrict__ dst) {
  const long sb = (long)C * H * W;
  const long sc = CL ? 1 : (long)H * W, sy = CL ? (long)W * C : W, sx = CL ? C : 1;
  const int bins = out_size * out_size;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c, bin, k;
    if (CL) { c = (int)(i % C); bin = (int)((i / C) % bins); k = (int)(i / ((long)C * bins)); }
    else { bin = (int)(i % bins); c = (int)((i / bins) % C); k = (int)(i / ((long)C * bins)); }
    const int ph = bin / out_size, pw = bin - ph * out_size;
    const RRoi g = rroi_geom(rois + (size_t)k * 6, out_size, scale, sample_num, aligned, clockwise, B);
    const long oidx = ((long)k * C + c) * bins + bin;            // [K,C,out,out]
    const long base = (long)g.b * sb + (long)c * sc;
    float acc = 0.f;
    const float gv = BWD ? src[oidx] / g.count : 0.f;
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const float yy = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.grid_h;
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const float xx = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.grid_w;
        const float y = yy * g.cosv - xx * g.sinv + g.ch;
        const float x = yy * g.sinv + xx * g.cosv + g.cw;
        const Tap4 t = tap4(y, x, H, W);
        if (!t.valid) continue;
        const long o1 = base + t.y0 * sy + t.x0 * sx, o2 = base + t.y0 * sy + t.x1 * sx;
        const long o3 = base + t.y1 * sy + t.x0 * sx, o4 = base + t.y1 * sy + t.x1 * sx;
        if (BWD) {
          atomicAdd(&dst[o1], gv * t.w1); atomicAdd(&dst[o2], gv * t.w2);
          atomicAdd(&dst[o3], gv * t.w3); atomicAdd(&dst[o4], gv * t.w4);
        } else {
          acc += t.w1 * src[o1] + t.w2 * src[o2] + t.w3 * src[o3] + t.w4 * src[o4];
        }
      }
    }
    if (!BWD) dst[oidx] = acc / g.count;
  }
}

// ---- channels_last fast path: one workgroup per RoI, one thread per channel --------------------
// The <= 196 sample taps of a RoI (49 bins x <= 4 samples; config: sample_num = 2) are computed ONCE by the
// first threads (one sin/cos per sample instead of one per output element) and shared through LDS.  The
// samples of a tiny oriented object revisit the same few feature pixels (footprint 9-36 px for 784 neighbour
// reads), so the footprint is staged on chip: [F][C] floats in LDS, thread c owning column c.
//   forward : footprint rows are read once (1 KiB coalesced each), the 784 neighbour reads hit LDS;
//   backward: gradients accumulate into the LDS footprint with plain read-modify-writes and are flushed with
//             ONE global atomic per footprint pixel and channel instead of four per sample.
// RoIs whose footprint exceeds RR_FMAX pixels read / scatter directly.
constexpr int RR_MAXS = 196;       // samples per RoI on the fast path
constexpr int RR_FMAX = 48;        // footprint pixels staged on chip

struct RTaps {
  int4 o[RR_MAXS];                 // neighbour index: y*W+x, rewritten to the footprint-local index when staged
  float4 w[RR_MAXS];               // bilinear weights / count (0 for samples outside the map)
  int y0, y1, x0, x1;              // bounding box of every neighbour
};

// Returns (block-uniform) the number of footprint pixels if the RoI is staged on chip, 0 if it is not, -1 if
// every sample lies outside the map.
__device__ __forceinline__ int rroi_taps(const RRoi& g, int out_size, int H, int W, RTaps* T) {
  if (threadIdx.x == 0) { T->y0 = 1 << 30; T->x0 = 1 << 30; T->y1 = -1; T->x1 = -1; }
  __syncthreads();
  const int per = g.grid_h * g.grid_w, n = out_size * out_size * per;
  int ly0 = 1 << 30, lx0 = 1 << 30, ly1 = -1, lx1 = -1;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int bin = t / per, sidx = t - bin * per;
    const int ph = bin / out_size, pw = bin - ph * out_size;
    const int iy = sidx / g.grid_w, ix = sidx - iy * g.grid_w;
    const float yy = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.grid_h;
    const float xx = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.grid_w;
    const float y = yy * g.cosv - xx * g.sinv + g.ch;
    const float x = yy * g.sinv + xx * g.cosv + g.cw;
    const Tap4 q = tap4(y, x, H, W);
    const float m = q.valid ? 1.f / g.count : 0.f;
    T->o[t] = make_int4((q.y0 << 16) | q.x0, (q.y0 << 16) | q.x1, (q.y1 << 16) | q.x0, (q.y1 << 16) | q.x1);
    T->w[t] = make_float4(q.w1 * m, q.w2 * m, q.w3 * m, q.w4 * m);
    if (q.valid) {
      ly0 = min(ly0, q.y0); ly1 = max(ly1, q.y1);
      lx0 = min(lx0, q.x0); lx1 = max(lx1, q.x1);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {                       // one LDS atomic per wave instead of one per sample
    ly0 = min(ly0, __shfl_xor(ly0, o, 64)); ly1 = max(ly1, __shfl_xor(ly1, o, 64));
    lx0 = min(lx0, __shfl_xor(lx0, o, 64)); lx1 = max(lx1, __shfl_xor(lx1, o, 64));
  }
  if ((threadIdx.x & 63) == 0 && ly1 >= 0) {
    atomicMin(&T->y0, ly0); atomicMax(&T->y1, ly1);
    atomicMin(&T->x0, lx0); atomicMax(&T->x1, lx1);
  }
  __syncthreads();
  if (T->y1 < T->y0) return -1;
  const int nx = T->x1 - T->x0 + 1, ny = T->y1 - T->y0 + 1;
  const bool staged = nx * ny <= RR_FMAX;
  const int y0 = T->y0, x0 = T->x0;
  for (int t = threadIdx.x; t < n * 4; t += blockDim.x) {          // packed (y,x) -> the index the main loop uses
    int* p = reinterpret_cast<int*>(&T->o[0]) + t;
    const int y = *p >> 16, x = *p & 0xffff;
    const bool dead = reinterpret_cast<const float*>(&T->w[0])[t] == 0.f;   // invalid sample or zero weight
    *p = dead ? 0 : (staged ? (y - y0) * nx + (x - x0) : y * W + x);
  }
  __syncthreads();
  return staged ? nx * ny : 0;
}

// Both kernels are specialised for out_size == 7: the 49 bin values of channel c live in 49 registers of thread
// c, and the [49][C+1] transpose tile (coalesced store of the output block / coalesced load of its gradient)
// shares the LDS region of the staged footprint - the two are never live at the same time.
constexpr int RR_BINS = 49;

template <int PER>
__global__ void __launch_bounds__(256)
    roi_align_rotated_fwd_cl(const float* __restrict__ feat, const float* __restrict__ rois, int B, int C, int H, int W,
                             float scale, int sample_num, int aligned, int clockwise, float* __restrict__ out) {
  extern __shared__ float fp[];    // [RR_FMAX][C] staged footprint, later the [49][C+1] output tile
  __shared__ RTaps T;
  const int k = blockIdx.x;
  const RRoi g = rroi_geom(rois + (size_t)k * 6, 7, scale, sample_num, aligned, clockwise, B);
  const int F = rroi_taps(g, 7, H, W, &T);
  const int ld = C + 1;
  const float* fb = feat + (size_t)g.b * H * W * C;
  for (int c0 = 0; c0 < C; c0 += blockDim.x) {
    const int c = c0 + threadIdx.x;
    float acc[RR_BINS];
#pragma unroll
    for (int bin = 0; bin < RR_BINS; ++bin) acc[bin] = 0.f;
    if (c < C && F >= 0) {
      if (F > 0) {
        const int nx = T.x1 - T.x0 + 1;
        for (int pix = 0; pix < F; ++pix)          // own column: no barrier between staging and use
          fp[pix * C + c] = fb[((size_t)(T.y0 + pix / nx) * W + T.x0 + pix % nx) * C + c];
      }
      if (F > 0) {              // block-uniform: hoisted so that each variant is one straight-line block
#pragma unroll
        for (int bin = 0; bin < RR_BINS; ++bin) {
          float a = 0.f;
#pragma unroll
          for (int sidx = 0; sidx < PER; ++sidx) {            // 16 independent LDS reads in flight per bin
            const int4 o = T.o[bin * PER + sidx];
            const float4 w = T.w[bin * PER + sidx];
            a += w.x * fp[o.x * C + c] + w.y * fp[o.y * C + c] + w.z * fp[o.z * C + c] + w.w * fp[o.w * C + c];
          }
          acc[bin] = a;
        }
      } else {
#pragma unroll
        for (int bin = 0; bin < RR_BINS; ++bin) {
          float a = 0.f;
#pragma unroll
          for (int sidx = 0; sidx < PER; ++sidx) {
            const int4 o = T.o[bin * PER + sidx];
            const float4 w = T.w[bin * PER + sidx];
            a += w.x * fb[(size_t)o.x * C + c] + w.y * fb[(size_t)o.y * C + c] + w.z * fb[(size_t)o.z * C + c] +
                 w.w * fb[(size_t)o.w * C + c];
          }
          acc[bin] = a;
        }
      }
    }
    __syncthreads();                               // every column has been consumed: the region becomes the tile
    if (c < C) {
#pragma unroll
      for (int bin = 0; bin < RR_BINS; ++bin) fp[bin * ld + c] = acc[bin];
    }
    __syncthreads();
    const int cn = min(C - c0, (int)blockDim.x);   // channels of this pass
    float* ob = out + ((size_t)k * C + c0) * RR_BINS;
    for (int o = threadIdx.x; o < cn * RR_BINS; o += blockDim.x) {
      const int cc = o / RR_BINS, bin = o - cc * RR_BINS;
      ob[o] = fp[bin * ld + c0 + cc];
    }
    __syncthreads();
  }
}

// Backward as a GATHER over the footprint.  A scatter (sample -> 4 pixels) into an on-chip accumulator
// serialises on LDS read-modify-write round trips because consecutive samples hit the same pixels, and LDS
// float atomics execute one lane per clock on this part (measured: 8x slower).  Instead the taps are
// bucketed by footprint pixel once per RoI (counting sort of <= 784 (bin, weight) entries, done by the first
// threads); thread c then walks the entry list of each footprint pixel, reads the gradient tile (coalesced
// load, transposed through LDS) with independent LDS reads, and issues ONE global atomic per footprint pixel
// and channel (9-36 instead of 784).  RoIs with a footprint above RR_FMAX pixels scatter directly.
struct RCsr {
  int start[RR_FMAX + 1];
  int cur[RR_FMAX];
  int bin[RR_MAXS * 4];
  float w[RR_MAXS * 4];
};

template <int PER>
__global__ void __launch_bounds__(256)
    roi_align_rotated_bwd_cl(const float* __restrict__ gout, const float* __restrict__ rois, int B, int C, int H, int W,
                             float scale, int sample_num, int aligned, int clockwise, float* __restrict__ gfeat) {
  extern __shared__ float fp[];    // the [49][C+1] gradient tile
  __shared__ RTaps T;
  __shared__ RCsr S;
  const int k = blockIdx.x;
  const RRoi g = rroi_geom(rois + (size_t)k * 6, 7, scale, sample_num, aligned, clockwise, B);
  const int F = rroi_taps(g, 7, H, W, &T);
  if (F < 0) return;                                        // every sample outside the map (block-uniform)
  const int ld = C + 1, NE = RR_BINS * PER * 4;
  float* fb = gfeat + (size_t)g.b * H * W * C;
  const int* To = reinterpret_cast<const int*>(&T.o[0]);
  const float* Tw = reinterpret_cast<const float*>(&T.w[0]);
  if (F > 0) {                                              // bucket the taps by footprint pixel
    for (int i = threadIdx.x; i <= F; i += blockDim.x) S.start[i] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < NE; e += blockDim.x)
      if (Tw[e] != 0.f) atomicAdd(&S.start[To[e] + 1], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
      int run = 0;
      for (int i = 0; i < F; ++i) { run += S.start[i + 1]; S.start[i + 1] = run; S.cur[i] = S.start[i]; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NE; e += blockDim.x) {
      const float w = Tw[e];
      if (w != 0.f) {
        const int pos = atomicAdd(&S.cur[To[e]], 1);
        S.bin[pos] = e / (PER * 4);
        S.w[pos] = w;
      }
    }
  }
  const int nx = T.x1 - T.x0 + 1;
  for (int c0 = 0; c0 < C; c0 += blockDim.x) {
    const int c = c0 + threadIdx.x;
    const int cn = min(C - c0, (int)blockDim.x);
    const float* gb = gout + ((size_t)k * C + c0) * RR_BINS;
    __syncthreads();
    for (int o = threadIdx.x; o < cn * RR_BINS; o += blockDim.x) {      // coalesced read, transposed through LDS
      const int cc = o / RR_BINS, bin = o - cc * RR_BINS;
      fp[bin * ld + c0 + cc] = gb[o];
    }
    __syncthreads();
    if (c >= C) continue;
    if (F > 0) {
      for (int pix = 0; pix < F; ++pix) {
        float a = 0.f;
        const int e1 = S.start[pix + 1];
        int e = S.start[pix];
        for (; e + 8 <= e1; e += 8) {                      // 8 independent (bin -> tile) read chains in flight
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = S.w[e + u] * fp[S.bin[e + u] * ld + c];
#pragma unroll
          for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; e < e1; ++e) a += S.w[e] * fp[S.bin[e] * ld + c];
        if (a != 0.f) atomicAdd(&fb[((size_t)(T.y0 + pix / nx) * W + T.x0 + pix % nx) * C + c], a);
      }
    } else {
      for (int e = 0; e < NE; ++e) {
        const float v = Tw[e] * fp[(e / (PER * 4)) * ld + c];
        if (v != 0.f) atomicAdd(&fb[(size_t)To[e] * C + c], v);
      }
    }
  }
}

// ---- r03: RoIAlignRotated as a small dense product per RoI ------------------------------------------------
// Bilinear pooling is linear: out[bin][c] = sum_p Wt[p][bin] * feat[p][c] over the F footprint pixels of the RoI, with
// Wt[p][bin] = the summed weights of the (<= 4 samples x 4 taps) of `bin` that land on pixel p, / count.  Wt is a [F <= 48][49]
// matrix per RoI; building it costs 784 scalar read-modify-writes ONCE per RoI, by 49 lanes that own one column each (no atomics,
// no sort).  After that every channel does F x 49 FMAs against broadcast ds_read_b128 rows of Wt:
//   forward : each footprint pixel value is read ONCE per RoI and channel (a coalesced 256-byte row per wave, straight from
//             L1 / L2 - nothing is staged), instead of 784 LDS reads + 392 tap reads per channel;
//   backward: grad_feat[p][c] = sum_bin Wt[p][bin] * g[bin][c] with the 49 gradients of the channel in registers - ONE global
//             atomic per footprint pixel and channel, no counting sort, no dependent LDS read-modify-write chains (the round-2
//             backward spent 1.4 ms per 5 000 RoIs mostly in the sort's same-address LDS atomics and its serial prefix sum).
// One 256-thread workgroup takes RR_GROUP = 4 consecutive RoIs: wave w builds the matrix of RoI w (the four builds run side by
// side), then every wave owns 64 channels of each RoI in turn.  The [K, C, 7, 7] block is written / read through a wave-private
// [32 channels][49] LDS tile: lane stride 49 words (conflict-free ds_write_b32 / ds_read_b32), 16-byte coalesced global accesses.
// LDS 65 KB -> 2 workgroups per CU.  RoIs with a footprint above RR_FMAX pixels (the large synthetic rectangles of burn-in step 1)
// keep their 196 samples' taps in the matrix's place and read / scatter per sample.
constexpr int RR_GROUP = 4;
constexpr int RR_WLD = 52;                         // row pitch of Wt in floats (49 bins, padded to 16-byte multiples)
constexpr int RR_WT_FLOATS = RR_FMAX * RR_WLD;     // 2 496 floats = 9 984 B per RoI; >= 196 * 8 floats of per-sample taps
static_assert(RR_WT_FLOATS >= RR_MAXS * 8, "the per-sample taps of a large RoI live in the matrix's place");

// Backward only: a footprint of 49 ... RR_CSRMAX pixels keeps the <= 784 (pixel, bin, weight) entries of the RoI SORTED BY PIXEL in
// the matrix's place (counting sort inside the building wavefront: per-pixel counts by LDS atomics, a wave-wide prefix sum, a
// scatter): every channel then adds its entries pixel by pixel and issues ONE atomic per touched pixel instead of four per sample
// (784 per channel and RoI) - phase 1's synthetic bags have footprints of 64 (median) ... 224 (p90) pixels.
constexpr int RR_CSRMAX = 448;                     // 64 lanes x 7 pixels of the prefix sum (capacity of the layout)
// Used up to RR_CSRUSE pixels (>= 6 entries per pixel): the route trades 784 fire-and-forget atomics for a chain of 784 dependent
// (LDS entry -> cached gradient -> FMA) steps, which only pays while the launch is bound by the atomic rate.  With the cap at 448
// phase 1's launch went 2.55 -> 1.30 ms but the benchmark's phase 2 (fewer, larger bags) lost 1.3 ms / iteration.
constexpr int RR_CSRUSE = 128;
static_assert(RR_WT_FLOATS >= (RR_CSRMAX + 1) + RR_CSRMAX + 2 * RR_MAXS * 4, "start | cursor | (bin, weight) entries fit the matrix area");

struct RMeta {                                     // per RoI, wave-uniform
  int F, y0, x0, nx, b;                            // F > 0: matrix path; F == 0: per-sample path; F == -1: nothing to do;
  int Fc;                                          // F == -2: pixel-sorted entries over Fc = nx * ny footprint pixels
};

// wave `w` of the workgroup prepares RoI k: geometry, footprint, Wt (or the per-sample taps)
template <bool CSR = false>
__device__ __forceinline__ void rroi_build(const float* __restrict__ roi, int B, int H, int W, float scale, int aligned, int clockwise,
                                           float* __restrict__ wt, int* __restrict__ poff, RMeta* __restrict__ meta) {
  const int lane = threadIdx.x & 63;
  const RRoi g = rroi_geom(roi, 7, scale, 2, aligned, clockwise, B);
  // lane = bin (< 49): its 2 x 2 samples
  Tap4 q[4];
  int ly0 = 1 << 30, lx0 = 1 << 30, ly1 = -1, lx1 = -1;
  const int ph = lane / 7, pw = lane - ph * 7;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const int iy = s4 >> 1, ix = s4 & 1;
    const float yy = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / 2.f;
    const float xx = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / 2.f;
    const float y = yy * g.cosv - xx * g.sinv + g.ch;
    const float x = yy * g.sinv + xx * g.cosv + g.cw;
    q[s4] = tap4(y, x, H, W);
    if (lane >= RR_BINS) q[s4].valid = false;
    if (q[s4].valid) {
      ly0 = min(ly0, q[s4].y0); ly1 = max(ly1, q[s4].y1);
      lx0 = min(lx0, q[s4].x0); lx1 = max(lx1, q[s4].x1);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ly0 = min(ly0, __shfl_xor(ly0, o, 64)); ly1 = max(ly1, __shfl_xor(ly1, o, 64));
    lx0 = min(lx0, __shfl_xor(lx0, o, 64)); lx1 = max(lx1, __shfl_xor(lx1, o, 64));
  }
  const int nx = lx1 - lx0 + 1, ny = ly1 - ly0 + 1;
  int F = ly1 < 0 ? -1 : (nx * ny <= RR_FMAX ? nx * ny : 0);
  if (CSR && F == 0 && nx * ny <= RR_CSRUSE) F = -2;
  if (lane == 0) { meta->F = F; meta->y0 = ly0; meta->x0 = lx0; meta->nx = nx; meta->b = g.b; meta->Fc = nx * ny; }
  const float inv = 1.f / g.count;
  // This lane (= bin) owns 16 (pixel, weight) entries, merged IN REGISTERS first (the 2 x 2 samples of a bin revisit the same pixels)
  int px[16];
  float pw_[16];
  if (F > 0 || (CSR && F == -2)) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int r0 = (q[s4].y0 - ly0) * nx - lx0, r1 = (q[s4].y1 - ly0) * nx - lx0;
      const float m = (q[s4].valid && lane < RR_BINS) ? inv : 0.f;
      px[4 * s4] = r0 + q[s4].x0; px[4 * s4 + 1] = r0 + q[s4].x1; px[4 * s4 + 2] = r1 + q[s4].x0; px[4 * s4 + 3] = r1 + q[s4].x1;
      pw_[4 * s4] = q[s4].w1 * m; pw_[4 * s4 + 1] = q[s4].w2 * m; pw_[4 * s4 + 2] = q[s4].w3 * m; pw_[4 * s4 + 3] = q[s4].w4 * m;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int j = i + 1; j < 16; ++j) {                      // fold a later duplicate into the earlier entry
        const bool same = px[j] == px[i];
        pw_[i] += same ? pw_[j] : 0.f;
        pw_[j] = same ? 0.f : pw_[j];
      }
  }
  if (F > 0) {
    for (int i = lane; i < F * (RR_WLD / 4); i += 64) reinterpret_cast<float4*>(wt)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < F) poff[lane] = (lane / nx) * W + lane % nx;
    // (same wave: LDS operations execute in order, the zeros land before the stores below.)  Each surviving entry is ONE plain
    // store into this lane's column of Wt: a chain of 16 dependent LDS read-modify-writes per lane cost ~2 000 cycles per RoI.
    if (lane < RR_BINS) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (pw_[i] != 0.f) wt[px[i] * RR_WLD + lane] = pw_[i];
    }
  } else if (CSR && F == -2) {                     // counting sort of the entries by pixel: start[Fc + 1] | cursor[Fc] | (bin, weight)[<= 784]
    const int Fc = nx * ny;
    int* start = reinterpret_cast<int*>(wt);
    int* cur = start + (RR_CSRMAX + 1);
    int* ent = cur + RR_CSRMAX;
    for (int i = lane; i < Fc; i += 64) { start[i] = 0; cur[i] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (pw_[i] != 0.f) atomicAdd(&start[px[i]], 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int cnt[7], sum = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int idx = 7 * lane + j;
      cnt[j] = idx < Fc ? start[idx] : 0;
      sum += cnt[j];
    }
    int incl = sum;                                  // inclusive wave scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    int run = incl - sum;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int idx = 7 * lane + j;
      if (idx < Fc) start[idx] = run;
      run += cnt[j];
    }
    if (lane == 63) start[Fc] = incl;                // the total (lane 63's pixels 441 .. 447 end at or before Fc)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (pw_[i] != 0.f) {
        const int pos = start[px[i]] + atomicAdd(&cur[px[i]], 1);
        ent[2 * pos] = lane;
        ent[2 * pos + 1] = __float_as_int(pw_[i]);
      }
  } else if (F == 0 && lane < RR_BINS) {           // per-sample taps: [bin][sample] x (4 offsets | 4 weights)
    int* to = reinterpret_cast<int*>(wt);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int e = (lane * 4 + s4) * 8;
      const float m = q[s4].valid ? inv : 0.f;
      to[e] = q[s4].y0 * W + q[s4].x0; to[e + 1] = q[s4].y0 * W + q[s4].x1;
      to[e + 2] = q[s4].y1 * W + q[s4].x0; to[e + 3] = q[s4].y1 * W + q[s4].x1;
      wt[e + 4] = q[s4].w1 * m; wt[e + 5] = q[s4].w2 * m; wt[e + 6] = q[s4].w3 * m; wt[e + 7] = q[s4].w4 * m;
    }
  }
}

struct RGroupSmem {
  float wt[RR_GROUP][RR_WT_FLOATS];
  float tile[4][32 * RR_BINS];                      // wave-private transpose tiles
  int off[RR_GROUP][RR_FMAX];                       // footprint pixel p -> (y * W + x) relative to the footprint's corner
  RMeta meta[RR_GROUP];
};

__global__ void __launch_bounds__(256)
    roi_align_rotated_fwd_mm(const float* __restrict__ feat, const float* __restrict__ rois, int B, int C, int H, int W, int K,
                             float scale, int aligned, int clockwise, int group, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rr_smem[];
  RGroupSmem& S = *reinterpret_cast<RGroupSmem*>(rr_smem);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k0 = blockIdx.x * group;
  if (w < group && k0 + w < K) rroi_build(rois + (size_t)(k0 + w) * 6, B, H, W, scale, aligned, clockwise, S.wt[w], S.off[w], &S.meta[w]);
  __syncthreads();
  float* tile = S.tile[w];
  for (int rr = 0; rr < group && k0 + rr < K; ++rr) {
    const RMeta m = S.meta[rr];
    const float* wt = S.wt[rr];
    const float* fb = feat + (size_t)m.b * H * W * C;
    for (int c0 = w * 64; c0 < C; c0 += 256) {               // this wave's 64 channels (C = 256: one pass)
      const int c = c0 + lane;
      const bool live = c < C;
      float acc[RR_BINS];
#pragma unroll
      for (int bin = 0; bin < RR_BINS; ++bin) acc[bin] = 0.f;
      if (m.F > 0) {
        const float* px = fb + ((size_t)m.y0 * W + m.x0) * C + (live ? c : 0);
        const int* po = S.off[rr];
        for (int p0 = 0; p0 < m.F; p0 += 8) {                 // 8 footprint rows (256 B per wave each) in flight at a time
          float fv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) fv[u] = (live && p0 + u < m.F) ? px[(size_t)po[p0 + u] * C] : 0.f;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (p0 + u < m.F) {                               // wave-uniform
              const float cur = fv[u];
              const float4* row = reinterpret_cast<const float4*>(wt + (p0 + u) * RR_WLD);
#pragma unroll
              for (int j = 0; j < 12; ++j) {
                const float4 w4 = row[j];
                acc[4 * j] = fmaf(w4.x, cur, acc[4 * j]); acc[4 * j + 1] = fmaf(w4.y, cur, acc[4 * j + 1]);
                acc[4 * j + 2] = fmaf(w4.z, cur, acc[4 * j + 2]); acc[4 * j + 3] = fmaf(w4.w, cur, acc[4 * j + 3]);
              }
              acc[48] = fmaf(wt[(p0 + u) * RR_WLD + 48], cur, acc[48]);
            }
          }
        }
      } else if (m.F == 0) {
        const int* to = reinterpret_cast<const int*>(wt);
        const float* fc = fb + (live ? c : 0);
#pragma unroll
        for (int bin = 0; bin < RR_BINS; ++bin) {
          float a = 0.f;
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const int e = (bin * 4 + s4) * 8;
            const int4 o = *reinterpret_cast<const int4*>(to + e);
            const float4 ww = *reinterpret_cast<const float4*>(wt + e + 4);
            if (live) a += ww.x * fc[(size_t)o.x * C] + ww.y * fc[(size_t)o.y * C] + ww.z * fc[(size_t)o.z * C] + ww.w * fc[(size_t)o.w * C];
          }
          acc[bin] = a;
        }
      }
      // [64 channels][49] -> global, 32 channels at a time through the wave's tile
      const int cn = min(C - c0, 64);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if ((lane >> 5) == hh) {
#pragma unroll
          for (int bin = 0; bin < RR_BINS; ++bin) tile[(lane & 31) * RR_BINS + bin] = acc[bin];
        }
        const int nch = min(max(cn - 32 * hh, 0), 32);        // channels of this half that exist
        float* ob = out + ((size_t)(k0 + rr) * C + c0 + 32 * hh) * RR_BINS;
        const int n4 = nch * RR_BINS / 4, rem = nch * RR_BINS - n4 * 4;   // (nch = 32: 392 float4, no remainder)
        for (int i = lane; i < n4; i += 64) reinterpret_cast<float4*>(ob)[i] = reinterpret_cast<const float4*>(tile)[i];
        if (lane < rem) ob[n4 * 4 + lane] = tile[n4 * 4 + lane];
      }
    }
  }
}

// CSR: the instantiation for bags (K >= 2 048: four RoIs per workgroup) carries the pixel-sorted route; the one for small batches
// (the 400 negatives: footprints of thousands of pixels, always on the per-sample route) does not - with the extra route compiled
// in, the per-sample route of that launch measured 40 % slower (register allocation / scheduling of the whole kernel).
template <bool CSR>
__global__ void __launch_bounds__(256)
    roi_align_rotated_bwd_mm(const float* __restrict__ gout, const float* __restrict__ rois, int B, int C, int H, int W, int K,
                             float scale, int aligned, int clockwise, int group, float* __restrict__ gfeat) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rr_smem[];
  RGroupSmem& S = *reinterpret_cast<RGroupSmem*>(rr_smem);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k0 = blockIdx.x * group;
  if (w < group && k0 + w < K) rroi_build<CSR>(rois + (size_t)(k0 + w) * 6, B, H, W, scale, aligned, clockwise, S.wt[w], S.off[w], &S.meta[w]);
  __syncthreads();
  float* tile = S.tile[w];
  for (int rr = 0; rr < group && k0 + rr < K; ++rr) {
    const RMeta m = S.meta[rr];
    if (m.F == -1) continue;
    const float* wt = S.wt[rr];
    float* fb = gfeat + (size_t)m.b * H * W * C;
    for (int c0 = w * 64; c0 < C; c0 += 256) {
      const int c = c0 + lane;
      const bool live = c < C;
      const int cn = min(C - c0, 64);
      if (CSR && m.F == -2) {                                 // pixel-sorted entries: one atomic per touched footprint pixel
        const int* start = reinterpret_cast<const int*>(S.wt[rr]);
        const int* ent = start + (RR_CSRMAX + 1) + RR_CSRMAX;
        const float* gl = gout + ((size_t)(k0 + rr) * C + (live ? c : 0)) * RR_BINS;   // this lane's 49 gradients (cache resident)
        float* px = gfeat + (size_t)m.b * H * W * C + ((size_t)m.y0 * W + m.x0) * C + (live ? c : 0);
        int x = 0, rowoff = 0;
        for (int p = 0; p < m.Fc; ++p) {
          const int e0 = start[p], e1 = start[p + 1];
          float a = 0.f;
          for (int e = e0; e < e1; ++e) a = fmaf(__int_as_float(ent[2 * e + 1]), gl[ent[2 * e]], a);
          if (live && e1 > e0 && a != 0.f) atomicAdd(px + (size_t)(rowoff + x) * C, a);
          if (++x == m.nx) { x = 0; rowoff += W; }
        }
        continue;
      }
      float g[RR_BINS];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {                        // the 49 gradients of this lane's channel, through the tile
        const int nch = min(max(cn - 32 * hh, 0), 32);
        const float* gb = gout + ((size_t)(k0 + rr) * C + c0 + 32 * hh) * RR_BINS;
        const int n4 = nch * RR_BINS / 4, rem = nch * RR_BINS - n4 * 4;
        for (int i = lane; i < n4; i += 64) reinterpret_cast<float4*>(tile)[i] = reinterpret_cast<const float4*>(gb)[i];
        if (lane < rem) tile[n4 * 4 + lane] = gb[n4 * 4 + lane];
        if ((lane >> 5) == hh) {
#pragma unroll
          for (int bin = 0; bin < RR_BINS; ++bin) g[bin] = live ? tile[(lane & 31) * RR_BINS + bin] : 0.f;
        }
      }
      if (m.F > 0) {
        float* px = fb + ((size_t)m.y0 * W + m.x0) * C + (live ? c : 0);
        const int* po = S.off[rr];
        for (int p = 0; p < m.F; ++p) {
          const float4* row = reinterpret_cast<const float4*>(wt + p * RR_WLD);
          float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
          for (int j = 0; j < 12; ++j) {
            const float4 w4 = row[j];
            a0 = fmaf(w4.x, g[4 * j], a0); a1 = fmaf(w4.y, g[4 * j + 1], a1);
            a2 = fmaf(w4.z, g[4 * j + 2], a2); a3 = fmaf(w4.w, g[4 * j + 3], a3);
          }
          const float a = (a0 + a1) + (a2 + a3) + wt[p * RR_WLD + 48] * g[48];
          if (live && a != 0.f) atomicAdd(px + (size_t)po[p] * C, a);
        }
      } else {
        const int* to = reinterpret_cast<const int*>(wt);
        float* fc = fb + (live ? c : 0);
#pragma unroll
        for (int bin = 0; bin < RR_BINS; ++bin) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const int e = (bin * 4 + s4) * 8;
            const int4 o = *reinterpret_cast<const int4*>(to + e);
            const float4 ww = *reinterpret_cast<const float4*>(wt + e + 4);
            const float gv = g[bin];
            if (live && gv != 0.f) {
              if (ww.x != 0.f) atomicAdd(fc + (size_t)o.x * C, gv * ww.x);
              if (ww.y != 0.f) atomicAdd(fc + (size_t)o.y * C, gv * ww.y);
              if (ww.z != 0.f) atomicAdd(fc + (size_t)o.z * C, gv * ww.z);
              if (ww.w != 0.f) atomicAdd(fc + (size_t)o.w * C, gv * ww.w);
            }
          }
        }
      }
    }
  }
}

}  // namespace pt

using namespace pt;

extern "C" int pt_diff_iou_rotated_fwd(const float* boxes1, const float* boxes2, int N, float* iou, void* stream) {
  if (N == 0) return PT_OK;
  PT_REQUIRE(boxes1 && boxes2 && iou && N > 0, PT_EINVAL, "pt_diff_iou_rotated_fwd: bad argument");
  hipLaunchKernelGGL(diff_iou_rotated_kernel, dim3(cdiv(N, 64)), dim3(64), 0, as_stream(stream), boxes1, boxes2, N,
                     (const float*)nullptr, iou, (float*)nullptr);
  PT_LAUNCH_CHECK("pt_diff_iou_rotated_fwd");
  return PT_OK;
}

extern "C" int pt_diff_iou_rotated_bwd(const float* boxes1, const float* boxes2, const float* grad_iou, int N,
                                       float* grad_boxes1, void* stream) {
  if (N == 0) return PT_OK;
  PT_REQUIRE(boxes1 && boxes2 && grad_iou && grad_boxes1 && N > 0, PT_EINVAL, "pt_diff_iou_rotated_bwd: bad argument");
  hipLaunchKernelGGL(diff_iou_rotated_kernel, dim3(cdiv(N, 64)), dim3(64), 0, as_stream(stream), boxes1, boxes2, N,
                     grad_iou, (float*)nullptr, grad_boxes1);
  PT_LAUNCH_CHECK("pt_diff_iou_rotated_bwd");
  return PT_OK;
}

template <bool BWD>
static int rroi_launch(const char* fn, const float* src, const float* rois, int B, int C, int H, int W, int K,
                       int out_size, float scale, int sample_num, int aligned, int clockwise, int channels_last,
                       float* dst, void* stream) {
  if (K == 0) return PT_OK;
  PT_REQUIRE(src && rois && dst && B > 0 && C > 0 && H > 0 && W > 0 && K > 0 && out_size >= 1, PT_EINVAL,
             "%s: bad argument", fn);
  const long total = (long)K * C * out_size * out_size;
  int nb = cdiv(total, 256);
  if (nb > 65536) nb = 65536;
  hipStream_t s = as_stream(stream);
  const size_t fpb = (size_t)RR_FMAX * C * sizeof(float), tlb = (size_t)RR_BINS * (C + 1) * sizeof(float);
  const size_t lds = fpb > tlb ? fpb : tlb;
  if (channels_last && out_size == 7 && sample_num == 2 && H * W < (1 << 24)) {
    // r03: four RoIs per workgroup, one small dense product per RoI (config 5: out_size 7, sample_num 2)
    // four RoIs per workgroup (their matrices are built side by side) once that still leaves >= 2 workgroups per CU; a small
    // batch (the 400 negatives: 70 x 70-pixel RoIs on the per-sample path) gets one workgroup per RoI
    const int group = K >= 2048 ? RR_GROUP : 1;
    static bool attr_mm[3] = {false, false, false};
    const int which = BWD ? (group > 1 ? 2 : 1) : 0;
    const void* kf = which == 2 ? reinterpret_cast<const void*>(roi_align_rotated_bwd_mm<true>)
                   : which == 1 ? reinterpret_cast<const void*>(roi_align_rotated_bwd_mm<false>)
                                : reinterpret_cast<const void*>(roi_align_rotated_fwd_mm);
    const size_t lds_mm = sizeof(RGroupSmem);
    if (!attr_mm[which]) {
      hipError_t e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mm);
      if (e != hipSuccess) { set_error("%s: LDS attribute: %s", fn, hipGetErrorString(e)); return (int)e; }
      attr_mm[which] = true;
    }
    if (which == 2)
      hipLaunchKernelGGL(roi_align_rotated_bwd_mm<true>, dim3(cdiv(K, group)), dim3(256), lds_mm, s, src, rois, B, C, H, W, K, scale,
                         aligned, clockwise, group, dst);
    else if (which == 1)
      hipLaunchKernelGGL(roi_align_rotated_bwd_mm<false>, dim3(cdiv(K, group)), dim3(256), lds_mm, s, src, rois, B, C, H, W, K, scale,
                         aligned, clockwise, group, dst);
    else
      hipLaunchKernelGGL(roi_align_rotated_fwd_mm, dim3(cdiv(K, group)), dim3(256), lds_mm, s, src, rois, B, C, H, W, K, scale, aligned,
                         clockwise, group, dst);
  } else if (channels_last && out_size == 7 && sample_num == 2 &&
      lds + sizeof(RTaps) <= 150 * 1024) {
    // fast path: one workgroup per RoI, 49 bins x 4 samples (config 5: out_size 7, sample_num 2)
    static size_t attr[2] = {0, 0};
    const void* kf = BWD ? reinterpret_cast<const void*>(roi_align_rotated_bwd_cl<4>)
                         : reinterpret_cast<const void*>(roi_align_rotated_fwd_cl<4>);
    if (lds > attr[BWD]) {
      hipError_t e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("%s: LDS attribute: %s", fn, hipGetErrorString(e)); return (int)e; }
      attr[BWD] = lds;
    }
    if (BWD)
      hipLaunchKernelGGL(roi_align_rotated_bwd_cl<4>, dim3(K), dim3(256), lds, s, src, rois, B, C, H, W, scale, sample_num,
                         aligned, clockwise, dst);
    else
      hipLaunchKernelGGL(roi_align_rotated_fwd_cl<4>, dim3(K), dim3(256), lds, s, src, rois, B, C, H, W, scale, sample_num,
                         aligned, clockwise, dst);
  } else if (channels_last)
    hipLaunchKernelGGL((roi_align_rotated_kernel<BWD, true>), dim3(nb), dim3(256), 0, s, src, rois, B, total, C, H, W,
                       out_size, scale, sample_num, aligned, clockwise, dst);
  else
    hipLaunchKernelGGL((roi_align_rotated_kernel<BWD, false>), dim3(nb), dim3(256), 0, s, src, rois, B, total, C, H, W,
                       out_size, scale, sample_num, aligned, clockwise, dst);
  PT_LAUNCH_CHECK(fn);
  return PT_OK;
}

extern "C" int pt_roi_align_rotated_fwd(const float* feat, const float* rois, int B, int C, int H, int W, int K,
                                        int out_size, float spatial_scale, int sample_num, int aligned, int clockwise,
                                        int channels_last, float* out, void* stream) {
  return rroi_launch<false>("pt_roi_align_rotated_fwd", feat, rois, B, C, H, W, K, out_size, spatial_scale, sample_num,
                            aligned, clockwise, channels_last, out, stream);
}

extern "C" int pt_roi_align_rotated_bwd(const float* grad_out, const float* rois, int B, int C, int H, int W, int K,
                                        int out_size, float spatial_scale, int sample_num, int aligned, int clockwise,
                                        int channels_last, float* grad_feat, void* stream) {
  return rroi_launch<true>("pt_roi_align_rotated_bwd", grad_out, rois, B, C, H, W, K, out_size, spatial_scale,
                           sample_num, aligned, clockwise, channels_last, grad_feat, stream);
}
