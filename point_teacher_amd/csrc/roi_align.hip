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

// Backward, channels_last.  One workgroup owns `group` CONSECUTIVE RoIs (the U2 jittered boxes
// of one MIL bag sit next to each other and cover the same few feature pixels).  Threads own
// channels, so the union footprint of the group (<= FOOT_MAXPIX pixels) is accumulated in LDS
// without atomics and flushed with ONE f32 atomic per (pixel, channel); with the separable
// weights a RoI issues (rows x cols of its footprint) updates instead of 4 per tap.  Groups
// whose union footprint is larger go straight to global atomics (256 contiguous bytes per
// wave instruction - the full-rate shape), still one per footprint pixel.
constexpr int FOOT_MAXPIX = 25;

__global__ void __launch_bounds__(256)
    roi_align_bwd_cl(const float* __restrict__ gout, const float* __restrict__ rois, int B, int C, int H, int W, int K,
                     int group, int out_size, float scale, int sampling_ratio, int aligned,
                     float* __restrict__ gfeat) {
  extern __shared__ float smem[];  // [bins][C+1] grad tile, [FOOT_MAXPIX][C] footprint, 8 ints of bounds
  __shared__ AxisW AX, AY;
  const int bins = out_size * out_size;
  const int ld = C + 1;
  float* tile = smem;
  float* foot = smem + (size_t)bins * ld;
  int* ub = reinterpret_cast<int*>(foot + (size_t)FOOT_MAXPIX * C);  // x0,y0,x1,y1 (inclusive), batch (-2 = mixed)
  const int k0 = blockIdx.x * group, k1 = min(k0 + group, K);
  if (threadIdx.x == 0) { ub[0] = 1 << 30; ub[1] = 1 << 30; ub[2] = -1; ub[3] = -1; ub[4] = -1; }
  __syncthreads();
  for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x) {
    const RoiGeom g = roi_geom(rois + (size_t)k * 5, out_size, scale, sampling_ratio, aligned, B);
    const float xe = g.start_w + g.bin_w * out_size, ye = g.start_h + g.bin_h * out_size;
    // taps of a sample at x touch floor(x) and floor(x)+1 (clamped); samples lie in [start, end]
    const int x0 = min(max((int)floorf(fminf(g.start_w, xe)), 0), W - 1);
    const int y0 = min(max((int)floorf(fminf(g.start_h, ye)), 0), H - 1);
    const int x1 = min(max((int)floorf(fmaxf(g.start_w, xe)) + 1, 0), W - 1);
    const int y1 = min(max((int)floorf(fmaxf(g.start_h, ye)) + 1, 0), H - 1);
    atomicMin(&ub[0], x0); atomicMin(&ub[1], y0); atomicMax(&ub[2], x1); atomicMax(&ub[3], y1);
    const int old = atomicCAS(&ub[4], -1, g.b);
    if (old != -1 && old != g.b) ub[4] = -2;
  }
  __syncthreads();
  const int ux0 = ub[0], uy0 = ub[1], uw = ub[2] - ub[0] + 1, uh = ub[3] - ub[1] + 1;
  const bool use_foot = (ub[4] >= 0) && (uw > 0) && (uh > 0) && (uw * uh <= FOOT_MAXPIX);
  if (use_foot)
    for (int i = threadIdx.x; i < uw * uh * C; i += blockDim.x) foot[i] = 0.f;
  for (int k = k0; k < k1; ++k) {
    const RoiGeom g = roi_geom(rois + (size_t)k * 5, out_size, scale, sampling_ratio, aligned, B);
    const float* gb = gout + (size_t)k * C * bins;
    __syncthreads();   // previous RoI's tile and weights fully consumed (and foot zeroed on the first trip)
    for (int o = threadIdx.x; o < C * bins; o += blockDim.x) {
      const int c = o / bins, bin = o - c * bins;
      tile[bin * ld + c] = gb[o];
    }
    axis_weights(g.start_w, g.bin_w, g.grid_w, W, out_size, &AX);   // ends with a barrier
    axis_weights(g.start_h, g.bin_h, g.grid_h, H, out_size, &AY);
    const int ny = AY.e - AY.o + 1, nx = AX.e - AX.o + 1;
    if (ny <= 0 || nx <= 0) continue;
    float* fb = gfeat + (size_t)g.b * H * W * C;
    const bool in = use_foot && AX.o >= ux0 && AX.e < ux0 + uw && AY.o >= uy0 && AY.e < uy0 + uh;
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
          if (v == 0.f) continue;
          if (in) {   // wave-uniform branch; thread-private column c of the footprint
            foot[((AY.o + py - uy0) * uw + (AX.o + px - ux0)) * C + c] += v;
          } else {
            atomicAdd(&fb[((size_t)(AY.o + py) * W + AX.o + px) * C + c], v);
          }
        }
      }
    }
  }
  if (use_foot) {
    // each thread flushes the columns it accumulated itself: no barrier needed
    float* fb = gfeat + (size_t)ub[4] * H * W * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x)
      for (int p = 0; p < uw * uh; ++p) {
        const float v = foot[p * C + c];
        if (v != 0.f) atomicAdd(&fb[((size_t)(uy0 + p / uw) * W + (ux0 + p % uw)) * C + c], v);
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
                                float spatial_scale, int sampling_ratio, int aligned, int channels_last, float* out,
                                void* stream) {
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

extern "C" int pt_roi_align_bwd(const float* grad_out, const float* rois, int B, int C, int H, int W, int K,
                                int out_size, float spatial_scale, int sampling_ratio, int aligned,
                                int channels_last, int group, float* grad_feat, void* stream) {
  if (K == 0) return PT_OK;
  int rc = roi_check("pt_roi_align_bwd", grad_out, rois, grad_feat, B, C, H, W, K, out_size, channels_last);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  if (channels_last) {
    const size_t lds = ((size_t)(C + 1) * out_size * out_size + (size_t)FOOT_MAXPIX * C + 8) * sizeof(float);
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
    hipLaunchKernelGGL(roi_align_bwd_cl, dim3(cdiv(K, group)), dim3(256), lds, s, grad_out, rois, B, C, H, W, K, group,
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
