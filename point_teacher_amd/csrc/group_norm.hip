// GroupNorm (+ ReLU) on channels_last activations for gfx950.
//
// The oriented config's dense head (OBB_TOD/mmrotate/models/dense_heads/rotated_fcos_head_p2rb_ts.py:135: norm_cfg=dict(type='GN',
// num_groups=32); towers = mmcv ConvModules built by mmdet's AnchorFreeHead, anchor_free_head.py:86-135) normalises every tower
// convolution's output with torch.nn.GroupNorm.  The
// library's kernels want NCHW: on the channels_last training layout every call paid two layout copies each way, handed the next
// tower convolution an NCHW tensor (so that one fell back to the library as well) and ran ~6 launches.  Here the activation stays
// [N, H*W, C] end to end:
//   forward   gn_stats_kernel   per (sample, 256-pixel chunk): sum and sum of squares of every group in float64 (a wavefront reads
//                               whole 1-KiB pixel rows) -> partials;
//             gn_apply_kernel   every workgroup folds the sample's partials (fixed order) into mean / rstd, then
//                               y = (x - mean) * rstd * gamma + beta (ReLU) for its chunk: x is read twice, y written once;
//   backward  gn_bwd_partial_kernel   per (sample, chunk, channel): A = sum dy, B = sum dy * xhat   (dy = gy * (y > 0));
//             gn_bwd_finalize_kernel  A, B per (sample, channel) -> dbeta, dgamma (sum over samples) and the two group sums
//                               s1 = sum gamma A, s2 = sum gamma B every element's gradient needs;
//             gn_bwd_apply_kernel     dx = rstd * (dy * gamma - (s1 + xhat * s2) / M),  M = H*W * C/G.
// All reductions run in a fixed order: results are bit-identical from run to run.  HBM-bound: forward 3, backward 7 passes over
// the activation.
#include "pt_common.h"

namespace pt {

constexpr int GN_PIX = 256;     // pixels per workgroup
constexpr int GN_MAXG = 64;     // groups per sample held in LDS

struct GnGeom {
  int HW, C, G, chunks;
  int quads, nph, qg;           // C / 4 channel quads, 256 / quads pixel phases per trip, quads per group
};

// partial sums of one chunk: ws[((n * chunks + ch) * G + g) * 2 + {0, 1}] (float64)
__global__ void __launch_bounds__(256) gn_stats_kernel(const float* __restrict__ x, GnGeom gg, double* __restrict__ ws) {
  __shared__ double red[256][2];
  const int n = blockIdx.y, ch = blockIdx.x;
  const int q = threadIdx.x % gg.quads, ph = threadIdx.x / gg.quads;
  const int p_end = min(gg.HW, (ch + 1) * GN_PIX);
  double s = 0.0, ss = 0.0;
  const float* base = x + ((size_t)n * gg.HW) * gg.C + 4 * q;
  for (int p = ch * GN_PIX + ph; p < p_end; p += gg.nph) {
    const float4 v = *reinterpret_cast<const float4*>(base + (size_t)p * gg.C);
    s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  red[threadIdx.x][0] = s;
  red[threadIdx.x][1] = ss;
  __syncthreads();
  if ((int)threadIdx.x < gg.G) {
    const int g = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int h = 0; h < gg.nph; ++h)
      for (int j = 0; j < gg.qg; ++j) {
        const int t = h * gg.quads + g * gg.qg + j;
        a += red[t][0];
        b += red[t][1];
      }
    double* o = ws + (((size_t)n * gg.chunks + ch) * gg.G + g) * 2;
    o[0] = a;
    o[1] = b;
  }
}

__global__ void __launch_bounds__(256)
    gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, GnGeom gg, float eps,
                    int relu, const double* __restrict__ ws, float* __restrict__ y, float* __restrict__ mean_out,
                    float* __restrict__ rstd_out) {
  __shared__ float sm[GN_MAXG], sr[GN_MAXG];
  const int n = blockIdx.y, ch = blockIdx.x;
  if ((int)threadIdx.x < gg.G) {
    const int g = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int c = 0; c < gg.chunks; ++c) {
      const double* o = ws + (((size_t)n * gg.chunks + c) * gg.G + g) * 2;
      a += o[0];
      b += o[1];
    }
    const double M = (double)gg.HW * (gg.C / gg.G);
    const double mu = a / M;
    double var = b / M - mu * mu;
    var = var > 0.0 ? var : 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    sm[g] = (float)mu;
    sr[g] = r;
    if (ch == 0) {
      mean_out[n * gg.G + g] = (float)mu;
      rstd_out[n * gg.G + g] = r;
    }
  }
  __syncthreads();
  const int q = threadIdx.x % gg.quads, ph = threadIdx.x / gg.quads;
  const int g = q / gg.qg;
  const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * q), be = *reinterpret_cast<const float4*>(beta + 4 * q);
  const float r = sr[g], mu = sm[g];
  const float a0 = r * ga.x, a1 = r * ga.y, a2 = r * ga.z, a3 = r * ga.w;
  const float b0 = be.x - mu * a0, b1 = be.y - mu * a1, b2 = be.z - mu * a2, b3 = be.w - mu * a3;
  const int p_end = min(gg.HW, (ch + 1) * GN_PIX);
  const size_t off = ((size_t)n * gg.HW) * gg.C + 4 * q;
  for (int p = ch * GN_PIX + ph; p < p_end; p += gg.nph) {
    const float4 v = *reinterpret_cast<const float4*>(x + off + (size_t)p * gg.C);
    float4 o = make_float4(fmaf(v.x, a0, b0), fmaf(v.y, a1, b1), fmaf(v.z, a2, b2), fmaf(v.w, a3, b3));
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *reinterpret_cast<float4*>(y + off + (size_t)p * gg.C) = o;
  }
}

// wsf[((n * chunks + ch) * C + c) * 2 + {0, 1}] = (sum dy, sum dy * xhat) of the chunk
__global__ void __launch_bounds__(256)
    gn_bwd_partial_kernel(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ y,
                          const float* __restrict__ mean, const float* __restrict__ rstd, GnGeom gg, float* __restrict__ wsf) {
  __shared__ float red[256][8];
  const int n = blockIdx.y, ch = blockIdx.x;
  const int q = threadIdx.x % gg.quads, ph = threadIdx.x / gg.quads;
  const int g = q / gg.qg;
  const float mu = mean[n * gg.G + g], r = rstd[n * gg.G + g];
  float A[4] = {0.f, 0.f, 0.f, 0.f}, Bv[4] = {0.f, 0.f, 0.f, 0.f};
  const int p_end = min(gg.HW, (ch + 1) * GN_PIX);
  const size_t off = ((size_t)n * gg.HW) * gg.C + 4 * q;
  for (int p = ch * GN_PIX + ph; p < p_end; p += gg.nph) {
    const size_t i = off + (size_t)p * gg.C;
    float4 d = *reinterpret_cast<const float4*>(gy + i);
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    if (y) {
      const float4 o = *reinterpret_cast<const float4*>(y + i);
      d.x = o.x > 0.f ? d.x : 0.f; d.y = o.y > 0.f ? d.y : 0.f; d.z = o.z > 0.f ? d.z : 0.f; d.w = o.w > 0.f ? d.w : 0.f;
    }
    A[0] += d.x; A[1] += d.y; A[2] += d.z; A[3] += d.w;
    Bv[0] = fmaf(d.x, (v.x - mu) * r, Bv[0]); Bv[1] = fmaf(d.y, (v.y - mu) * r, Bv[1]);
    Bv[2] = fmaf(d.z, (v.z - mu) * r, Bv[2]); Bv[3] = fmaf(d.w, (v.w - mu) * r, Bv[3]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[threadIdx.x][j] = A[j]; red[threadIdx.x][4 + j] = Bv[j]; }
  __syncthreads();
  if (ph == 0) {                                       // phases 0 .. nph-1 of this quad, fixed order
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < gg.nph; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] += red[h * gg.quads + q][j]; b[j] += red[h * gg.quads + q][4 + j]; }
    float* o = wsf + (((size_t)n * gg.chunks + ch) * gg.C + 4 * q) * 2;
    *reinterpret_cast<float4*>(o) = make_float4(a[0], b[0], a[1], b[1]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(a[2], b[2], a[3], b[3]);
  }
}

// one thread per channel (grid = C / 256 rounded up); loops over the samples: group sums per sample, dgamma / dbeta over samples
__global__ void __launch_bounds__(256)
    gn_bwd_finalize_kernel(const float* __restrict__ wsf, const float* __restrict__ gamma, int N, GnGeom gg, float* __restrict__ gs,
                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float sa[256], sb[256];
  const int c = blockIdx.x * 256 + threadIdx.x;
  const bool act = c < gg.C;
  const float gm = act ? gamma[c] : 0.f;
  const int cg = gg.C / gg.G;                          // channels per group (a multiple of 4, divides 256 or is a multiple of it)
  float dg = 0.f, db = 0.f;
  for (int n = 0; n < N; ++n) {
    float a = 0.f, b = 0.f;
    if (act)
      for (int ch = 0; ch < gg.chunks; ++ch) {
        const float2 v = *reinterpret_cast<const float2*>(wsf + (((size_t)n * gg.chunks + ch) * gg.C + c) * 2);
        a += v.x;
        b += v.y;
      }
    db += a;
    dg += b;
    sa[threadIdx.x] = gm * a;
    sb[threadIdx.x] = gm * b;
    __syncthreads();
    if (act && cg <= 256 && (c % cg) == 0) {           // first channel of a group that lies inside this block
      float s1 = 0.f, s2 = 0.f;
      for (int j = 0; j < cg; ++j) { s1 += sa[threadIdx.x + j]; s2 += sb[threadIdx.x + j]; }
      gs[((size_t)n * gg.G + c / cg) * 2] = s1;
      gs[((size_t)n * gg.G + c / cg) * 2 + 1] = s2;
    }
    __syncthreads();
  }
  if (act) { dgamma[c] = dg; dbeta[c] = db; }
}

__global__ void __launch_bounds__(256)
    gn_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ y,
                        const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                        const float* __restrict__ gs, GnGeom gg, float* __restrict__ gx) {
  const int n = blockIdx.y, ch = blockIdx.x;
  const int q = threadIdx.x % gg.quads, ph = threadIdx.x / gg.quads;
  const int g = q / gg.qg;
  const float mu = mean[n * gg.G + g], r = rstd[n * gg.G + g];
  const float invM = 1.f / ((float)gg.HW * (float)(gg.C / gg.G));
  const float s1 = gs[((size_t)n * gg.G + g) * 2] * invM, s2 = gs[((size_t)n * gg.G + g) * 2 + 1] * invM;
  const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * q);
  const int p_end = min(gg.HW, (ch + 1) * GN_PIX);
  const size_t off = ((size_t)n * gg.HW) * gg.C + 4 * q;
  for (int p = ch * GN_PIX + ph; p < p_end; p += gg.nph) {
    const size_t i = off + (size_t)p * gg.C;
    float4 d = *reinterpret_cast<const float4*>(gy + i);
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    if (y) {
      const float4 o = *reinterpret_cast<const float4*>(y + i);
      d.x = o.x > 0.f ? d.x : 0.f; d.y = o.y > 0.f ? d.y : 0.f; d.z = o.z > 0.f ? d.z : 0.f; d.w = o.w > 0.f ? d.w : 0.f;
    }
    float4 o;
    o.x = r * (d.x * ga.x - (s1 + (v.x - mu) * r * s2));
    o.y = r * (d.y * ga.y - (s1 + (v.y - mu) * r * s2));
    o.z = r * (d.z * ga.z - (s1 + (v.z - mu) * r * s2));
    o.w = r * (d.w * ga.w - (s1 + (v.w - mu) * r * s2));
    *reinterpret_cast<float4*>(gx + i) = o;
  }
}

static int gn_geom(const char* fn, int N, int HW, int C, int G, GnGeom* gg) {
  PT_REQUIRE(N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, PT_EINVAL, "%s: bad size", fn);
  const int cg = C / G;
  PT_REQUIRE(C % 4 == 0 && cg % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && G <= GN_MAXG, PT_ELIMIT,
             "%s: C=%d G=%d outside the supported shapes (C/4 divides 256, groups of a multiple of 4 channels, G <= %d)", fn, C, G,
             GN_MAXG);
  PT_REQUIRE(cg <= 256 && 256 % cg == 0, PT_ELIMIT, "%s: %d channels per group must divide 256", fn, cg);
  *gg = GnGeom{HW, C, G, cdiv(HW, GN_PIX), C / 4, 256 / (C / 4), cg / 4};
  return PT_OK;
}

}  // namespace pt

using namespace pt;

extern "C" int64_t pt_group_norm_cl_workspace_bytes(int N, int HW, int C, int G) {
  if (N <= 0 || HW <= 0 || C <= 0 || G <= 0) return 0;
  const int64_t chunks = (HW + GN_PIX - 1) / GN_PIX;
  const int64_t fwd = (int64_t)N * chunks * G * 2 * 8;                       // float64 group partials
  const int64_t bwd = ((int64_t)N * chunks * C * 2 + (int64_t)N * G * 2) * 4;   // float32 channel partials + group sums
  return fwd > bwd ? fwd : bwd;
}

extern "C" int pt_group_norm_cl_fwd(const float* x, const float* gamma, const float* beta, int N, int HW, int C, int G, float eps,
                                    int relu, float* y, float* mean, float* rstd, void* workspace, void* stream) {
  PT_REQUIRE(x && gamma && beta && y && mean && rstd && workspace, PT_EINVAL, "pt_group_norm_cl_fwd: NULL pointer");
  GnGeom gg;
  int rc = gn_geom("pt_group_norm_cl_fwd", N, HW, C, G, &gg);
  if (rc) return rc;
  PT_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)workspace)) & 15) == 0, PT_EINVAL,
             "pt_group_norm_cl_fwd: buffers must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(gg.chunks, N), dim3(256), 0, s, x, gg, reinterpret_cast<double*>(workspace));
  PT_LAUNCH_CHECK("pt_group_norm_cl_fwd (stats)");
  hipLaunchKernelGGL(gn_apply_kernel, dim3(gg.chunks, N), dim3(256), 0, s, x, gamma, beta, gg, eps, relu,
                     reinterpret_cast<const double*>(workspace), y, mean, rstd);
  PT_LAUNCH_CHECK("pt_group_norm_cl_fwd");
  return PT_OK;
}

extern "C" int pt_group_norm_cl_bwd(const float* grad_y, const float* x, const float* y, const float* gamma, const float* mean,
                                    const float* rstd, int N, int HW, int C, int G, float* grad_x, float* grad_gamma,
                                    float* grad_beta, void* workspace, void* stream) {
  PT_REQUIRE(grad_y && x && gamma && mean && rstd && grad_x && grad_gamma && grad_beta && workspace, PT_EINVAL,
             "pt_group_norm_cl_bwd: NULL pointer");
  GnGeom gg;
  int rc = gn_geom("pt_group_norm_cl_bwd", N, HW, C, G, &gg);
  if (rc) return rc;
  PT_REQUIRE(((((uintptr_t)grad_y) | ((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)grad_x) | ((uintptr_t)gamma) | ((uintptr_t)workspace)) & 15) == 0,
             PT_EINVAL, "pt_group_norm_cl_bwd: buffers must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  float* wsf = reinterpret_cast<float*>(workspace);
  float* gs = wsf + (size_t)N * gg.chunks * C * 2;
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(gg.chunks, N), dim3(256), 0, s, grad_y, x, y, mean, rstd, gg, wsf);
  PT_LAUNCH_CHECK("pt_group_norm_cl_bwd (partials)");
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, wsf, gamma, N, gg, gs, grad_gamma, grad_beta);
  PT_LAUNCH_CHECK("pt_group_norm_cl_bwd (finalize)");
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(gg.chunks, N), dim3(256), 0, s, grad_y, x, y, gamma, mean, rstd, gs, gg, grad_x);
  PT_LAUNCH_CHECK("pt_group_norm_cl_bwd");
  return PT_OK;
}
