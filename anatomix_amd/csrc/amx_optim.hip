// anatomix_amd -- the optimizer of the contrastive step (pretraining/models/supcl_model.py:510-516, 584-590: torch.optim.AdamW
// over netG and over netF; stepped at supcl_model.py:628-661).  ONE launch updates every parameter tensor of an optimizer:
// the tensor descriptors travel in the kernel arguments (so a captured HIP graph replays them as they are), a block finds its
// (tensor, chunk) from a prefix table with a scalar loop.  The arithmetic follows torch's AdamW term by term:
//     p *= 1 - lr wd;  m += (g - m)(1 - b1);  v = v b2 + (1 - b2) g g;
//     p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with the step count t read from the device (one fp32 scalar per tensor, as torch's capturable state keeps it).
#include <hip/hip_runtime.h>

#include "amx_common.h"

namespace amx {

constexpr int kAdamTensors = 48;             // descriptors per launch: 48 x 48 B + the prefix table < the 4 KiB of kernel arguments
constexpr int kAdamChunk = 4096;             // elements per block: 256 threads x 4 x float4

struct AdamArgs {
  float* p[kAdamTensors];
  const float* g[kAdamTensors];
  float* m[kAdamTensors];
  float* v[kAdamTensors];
  const float* step[kAdamTensors];
  long long n[kAdamTensors];
  int blk0[kAdamTensors + 1];
  int count;
  int maximize;
  double lr, b1, b2;                         // for the bias corrections (torch forms them in double on the host)
  float decay, w1, b2f, w2, eps;             // 1 - lr wd, 1 - b1, b2, 1 - b2: formed in double, then rounded once, as torch's scalars are
  // non-null: {lr, beta1, beta2, eps, weight_decay} are READ FROM THE DEVICE at launch time instead of the values above -- a step
  // captured in a HIP graph then follows an lr schedule (the reference changes lr every epoch: base_model.py update_learning_rate)
  // through a captured host-to-device copy of five doubles, instead of replaying the lr that was current at capture
  const double* hyper;
};

struct AdamCoef {
  float decay, w1, b1, b2, w2, inv_bc2s, eps, step_size, gsign;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamCoef& c) {
  g *= c.gsign;
  p *= c.decay;
  m = c.w1 < 0.5f ? m + c.w1 * (g - m) : g - (g - m) * c.b1;       // torch's lerp(m, g, 1 - b1), both of its branches
  v = v * c.b2 + c.w2 * g * g;
  const float denom = sqrtf(v) * c.inv_bc2s + c.eps;
  p = p - c.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  int t = 0;
  while (t + 1 < a.count && (int)blockIdx.x >= a.blk0[t + 1]) ++t;           // uniform: scalar loads from the argument segment
  const long long n = a.n[t], base = (long long)((int)blockIdx.x - a.blk0[t]) * kAdamChunk;
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const double step = (double)*a.step[t];
  double lr = a.lr, b1 = a.b1, b2 = a.b2;
  AdamCoef c;
  c.decay = a.decay;
  c.w1 = a.w1;
  c.b2 = a.b2f;
  c.w2 = a.w2;
  c.eps = a.eps;
  if (a.hyper) {                               // same expressions as launch_adamw forms on the host, in double, rounded once
    lr = a.hyper[0]; b1 = a.hyper[1]; b2 = a.hyper[2];
    c.eps = (float)a.hyper[3];
    c.decay = (float)(1.0 - lr * a.hyper[4]);
    c.w1 = (float)(1.0 - b1);
    c.b2 = (float)b2;
    c.w2 = (float)(1.0 - b2);
  }
  const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
  c.b1 = 1.f - c.w1;
  c.inv_bc2s = (float)(1.0 / sqrt(bc2));
  c.step_size = (float)(lr / bc1);
  c.gsign = a.maximize ? -1.f : 1.f;
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  if (vec && base + kAdamChunk <= n) {
    float4 P[4], G[4], M[4], V[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long i = base + (k * 256 + threadIdx.x) * 4;
      P[k] = *(const float4*)(p + i); G[k] = *(const float4*)(g + i); M[k] = *(const float4*)(m + i); V[k] = *(const float4*)(v + i);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      adam_one(P[k].x, G[k].x, M[k].x, V[k].x, c);
      adam_one(P[k].y, G[k].y, M[k].y, V[k].y, c);
      adam_one(P[k].z, G[k].z, M[k].z, V[k].z, c);
      adam_one(P[k].w, G[k].w, M[k].w, V[k].w, c);
      const long long i = base + (k * 256 + threadIdx.x) * 4;
      *(float4*)(p + i) = P[k]; *(float4*)(m + i) = M[k]; *(float4*)(v + i) = V[k];
    }
    return;
  }
  const long long end = base + kAdamChunk < n ? base + kAdamChunk : n;
  for (long long i = base + threadIdx.x; i < end; i += 256) {
    float P = p[i], M = m[i], V = v[i];
    adam_one(P, g[i], M, V, c);
    p[i] = P; m[i] = M; v[i] = V;
  }
}

// table: count rows of 6 x 64-bit {p, g, m, v, step, numel} on the HOST
hipError_t launch_adamw(const long long* table, int count, double lr, double b1, double b2, double eps, double wd, int maximize,
                        hipStream_t st, const double* d_hyper) {
  for (int t0 = 0; t0 < count; t0 += kAdamTensors) {
    AdamArgs a;
    const int c = count - t0 < kAdamTensors ? count - t0 : kAdamTensors;
    long long blocks = 0;
    for (int t = 0; t < c; ++t) {
      const long long* r = table + (size_t)(t0 + t) * 6;
      a.p[t] = (float*)r[0]; a.g[t] = (const float*)r[1]; a.m[t] = (float*)r[2]; a.v[t] = (float*)r[3];
      a.step[t] = (const float*)r[4]; a.n[t] = r[5];
      a.blk0[t] = (int)blocks;
      blocks += (r[5] + kAdamChunk - 1) / kAdamChunk;
      if (blocks > 0x3fffffff) return hipErrorInvalidValue;
    }
    for (int t = c; t < kAdamTensors; ++t) {
      a.p[t] = nullptr; a.g[t] = nullptr; a.m[t] = nullptr; a.v[t] = nullptr; a.step[t] = nullptr; a.n[t] = 0;
    }
    for (int t = c; t <= kAdamTensors; ++t) a.blk0[t] = (int)blocks;
    a.count = c; a.maximize = maximize; a.lr = lr; a.b1 = b1; a.b2 = b2; a.hyper = d_hyper;
    a.decay = (float)(1.0 - lr * wd); a.w1 = (float)(1.0 - b1); a.b2f = (float)b2; a.w2 = (float)(1.0 - b2); a.eps = (float)eps;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace amx
