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
template <int CPT>  // channels per thread (C = 256 * CPT / ... ) handled by looping
__global__ void __launch_bounds__(256)
    roi_align_fwd_cl(const float* __restrict__ feat, const float* __restrict__ rois, int B, int C, int H, int W,
                     int out_size, float scale, int sampling_ratio, int aligned, float* __restrict__ out) {
  extern __shared__ float tile[];  // [bins][C+1]
  const int k = blockIdx.x;
  const RoiGeom g = roi_geom(rois + (size_t)k * 5, out_size, scale, sampling_ratio, aligned, B);
  const int bins = out_size * out_size;
  const int ld = C + 1;
  const float* fb = feat + (size_t)g.b * H * W * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    for (int ph = 0; ph < out_size; ++ph) {
      for (int pw = 0; pw < out_size; ++pw) {
        float acc = 0.f;
        for (int iy = 0; iy < g.grid_h; ++iy) {
          const float y = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.grid_h;
          for (int ix = 0; ix < g.grid_w; ++ix) {
            const float x = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.grid_w;
            const Bilin q = bilin(y, x, H, W);
            if (q.valid) {
              const float v1 = fb[((size_t)q.y0 * W + q.x0) * C + c];
              const float v2 = fb[((size_t)q.y0 * W + q.x1) * C + c];
              const float v3 = fb[((size_t)q.y1 * W + q.x0) * C + c];
              const float v4 = fb[((size_t)q.y1 * W + q.x1) * C + c];
              acc += q.w1 * v1 + q.w2 * v2 + q.w3 * v3 + q.w4 * v4;
            }
          }
        }
        tile[(ph * out_size + pw) * ld + c] = acc / g.inv_count;
      }
    }
  }
  __syncthreads();
  float* ob = out + (size_t)k * C * bins;
  for (int o = threadIdx.x; o < C * bins; o += blockDim.x) {
    const int c = o / bins, bin = o - c * bins;
    ob[o] = tile[bin * ld + c];
  }
}

// Backward, channels_last.  One workgroup owns `group` CONSECUTIVE RoIs (the U2 jittered boxes
// of one MIL bag sit next to each other and cover the same few feature pixels).  Threads own
// channels, so the union footprint of the group (<= FOOT_MAXPIX pixels) is accumulated in LDS
// without any atomics and flushed with ONE f32 atomic per (pixel, channel): ~300x fewer global
// atomics than scattering every bilinear tap, and no same-address serialisation between the
// members of a bag.  Groups whose union footprint is larger fall back to per-tap atomics.
constexpr int FOOT_MAXPIX = 25;

__global__ void __launch_bounds__(256)
    roi_align_bwd_cl(const float* __restrict__ gout, const float* __restrict__ rois, int B, int C, int H, int W, int K,
                     int group, int out_size, float scale, int sampling_ratio, int aligned,
                     float* __restrict__ gfeat) {
  extern __shared__ float smem[];  // [bins][C+1] grad tile, [FOOT_MAXPIX][C] footprint, 8 ints of bounds
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
    __syncthreads();   // previous RoI's tile fully consumed (and foot zeroed on the first trip)
    for (int o = threadIdx.x; o < C * bins; o += blockDim.x) {
      const int c = o / bins, bin = o - c * bins;
      tile[bin * ld + c] = gb[o];
    }
    __syncthreads();
    float* fb = gfeat + (size_t)g.b * H * W * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      for (int ph = 0; ph < out_size; ++ph) {
        for (int pw = 0; pw < out_size; ++pw) {
          const float gv = tile[(ph * out_size + pw) * ld + c] / g.inv_count;
          for (int iy = 0; iy < g.grid_h; ++iy) {
            const float y = g.start_h + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ++ix) {
              const float x = g.start_w + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.grid_w;
              const Bilin q = bilin(y, x, H, W);
              if (!q.valid) continue;
              const bool in = use_foot && q.x0 >= ux0 && q.x1 < ux0 + uw && q.y0 >= uy0 && q.y1 < uy0 + uh;
              if (in) {   // wave-uniform branch; thread-private column c of the footprint
                const int r0 = (q.y0 - uy0) * uw, r1 = (q.y1 - uy0) * uw;
                foot[(r0 + q.x0 - ux0) * C + c] += gv * q.w1;
                foot[(r0 + q.x1 - ux0) * C + c] += gv * q.w2;
                foot[(r1 + q.x0 - ux0) * C + c] += gv * q.w3;
                foot[(r1 + q.x1 - ux0) * C + c] += gv * q.w4;
              } else {
                atomicAdd(&fb[((size_t)q.y0 * W + q.x0) * C + c], gv * q.w1);
                atomicAdd(&fb[((size_t)q.y0 * W + q.x1) * C + c], gv * q.w2);
                atomicAdd(&fb[((size_t)q.y1 * W + q.x0) * C + c], gv * q.w3);
                atomicAdd(&fb[((size_t)q.y1 * W + q.x1) * C + c], gv * q.w4);
              }
            }
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
  if (channels_last)
    PT_REQUIRE((size_t)(C + 1) * out_size * out_size * 4 <= 160 * 1024, PT_ELIMIT,
               "%s: C=%d too large for the LDS tile", fn, C);
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
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_cl<1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("pt_roi_align_fwd: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
      attr_bytes = lds;
    }
    hipLaunchKernelGGL(roi_align_fwd_cl<1>, dim3(K), dim3(256), lds, s, feat, rois, B, C, H, W, out_size, spatial_scale,
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
    PT_REQUIRE(lds <= 160 * 1024, PT_ELIMIT, "pt_roi_align_bwd: C=%d too large for the LDS tiles", C);
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
