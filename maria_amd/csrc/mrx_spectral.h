// Device code shared by the spectral generators (screens, detector noise):
// Philox-4x32-10, Box-Muller, and the in-LDS Stockham inverse FFT.
#pragma once

#include "mrx_internal.h"

namespace mrx_dev {

constexpr int kBlock = 256;

// ---- Philox-4x32-10 (Salmon et al., SC'11) ---------------------------------
struct U4 {
  uint32_t x, y, z, w;
};

__host__ __device__ inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c.x;
    const uint64_t p1 = (uint64_t)M1 * c.z;
    U4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

// Box-Muller on two 32-bit words: a pair of independent standard normals.  The
// hardware transcendentals (v_log_f32 = log2, v_sin/v_cos_f32 take revolutions)
// are accurate to ~1e-6, far below what a noise draw needs.
__device__ __forceinline__ float2 box_muller(uint32_t a, uint32_t b) {
  const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);           // [0,1)
  const float rad = sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln u1
  return make_float2(rad * __builtin_amdgcn_cosf(u2), rad * __builtin_amdgcn_sinf(u2));
}

// ---- in-LDS Stockham inverse FFT, radix 4 (+ one radix-2 stage) ------------
// Autosort (natural order in and out, no bit reversal), ping-pong between two
// LDS images of n complex values.  tw[m] = exp(+2 pi i m / n) for m < n/4; the
// other twiddles follow from w^2 = w*w, w^3 = w^2*w and exp(i(a + pi/2)) = i exp(ia).
// Unnormalised (numpy.fft.ifft * n).  Returns the image that holds the result.
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__device__ __forceinline__ float2* fft_lds_inverse(float2* a, float2* b,
                                                   const float2* tw, int n,
                                                   int log2n) {
  float2* in = a;
  float2* out = b;
  int ns = 1, s = 0;
  const int q = n >> 2;
  for (; s + 2 <= log2n; s += 2, ns <<= 2) {
    const int tstride = q / ns;  // n / (4 ns)
    for (int j = threadIdx.x; j < q; j += kBlock) {
      const int k = j & (ns - 1);
      float2 v0 = in[j], v1 = in[j + q], v2 = in[j + 2 * q], v3 = in[j + 3 * q];
      if (ns > 1) {
        const float2 w1 = tw[k * tstride];
        const float2 w2 = cmul(w1, w1);
        const float2 w3 = cmul(w2, w1);
        v1 = cmul(v1, w1);
        v2 = cmul(v2, w2);
        v3 = cmul(v3, w3);
      }
      const float2 t0 = make_float2(v0.x + v2.x, v0.y + v2.y);
      const float2 t1 = make_float2(v0.x - v2.x, v0.y - v2.y);
      const float2 t2 = make_float2(v1.x + v3.x, v1.y + v3.y);
      const float2 t3 = make_float2(-(v1.y - v3.y), v1.x - v3.x);  // +i (v1 - v3)
      const int j0 = ((j - k) << 2) + k;
      out[j0] = make_float2(t0.x + t2.x, t0.y + t2.y);
      out[j0 + ns] = make_float2(t1.x + t3.x, t1.y + t3.y);
      out[j0 + 2 * ns] = make_float2(t0.x - t2.x, t0.y - t2.y);
      out[j0 + 3 * ns] = make_float2(t1.x - t3.x, t1.y - t3.y);
    }
    __syncthreads();
    float2* t = in;
    in = out;
    out = t;
  }
  if (s < log2n) {  // one radix-2 stage left (odd log2 n): ns == n/2
    const int h = n >> 1;
    for (int j = threadIdx.x; j < h; j += kBlock) {
      // twiddle exp(2 pi i j / n), j < n/2
      float2 w = tw[j & (q - 1)];
      if (j >= q) w = make_float2(-w.y, w.x);
      const float2 v0 = in[j];
      const float2 v1 = cmul(in[j + h], w);
      out[j] = make_float2(v0.x + v1.x, v0.y + v1.y);
      out[j + h] = make_float2(v0.x - v1.x, v0.y - v1.y);
    }
    __syncthreads();
    float2* t = in;
    in = out;
    out = t;
  }
  return in;
}

// The same transform on 2^lj interleaved sequences (element i of sequence b at
// index (i << lj) + b): the Stockham recursion with an initial stride.  n >= 4.
__device__ __forceinline__ float2* fft_lds_inverse_batched(float2* a, float2* b,
                                                           const float2* tw, int n,
                                                           int log2n, int lj) {
  float2* in = a;
  float2* out = b;
  int ns = 1, s = 0;
  const int q = n >> 2;
  const int bmask = (1 << lj) - 1;
  for (; s + 2 <= log2n; s += 2, ns <<= 2) {
    const int tstride = q / ns;
    for (int jj = threadIdx.x; jj < (q << lj); jj += kBlock) {
      const int j = jj >> lj, bb = jj & bmask;
      const int k = j & (ns - 1);
      float2 v0 = in[(j << lj) + bb], v1 = in[((j + q) << lj) + bb];
      float2 v2 = in[((j + 2 * q) << lj) + bb], v3 = in[((j + 3 * q) << lj) + bb];
      if (ns > 1) {
        const float2 w1 = tw[k * tstride];
        const float2 w2 = cmul(w1, w1);
        const float2 w3 = cmul(w2, w1);
        v1 = cmul(v1, w1);
        v2 = cmul(v2, w2);
        v3 = cmul(v3, w3);
      }
      const float2 t0 = make_float2(v0.x + v2.x, v0.y + v2.y);
      const float2 t1 = make_float2(v0.x - v2.x, v0.y - v2.y);
      const float2 t2 = make_float2(v1.x + v3.x, v1.y + v3.y);
      const float2 t3 = make_float2(-(v1.y - v3.y), v1.x - v3.x);
      const int j0 = ((j - k) << 2) + k;
      out[(j0 << lj) + bb] = make_float2(t0.x + t2.x, t0.y + t2.y);
      out[((j0 + ns) << lj) + bb] = make_float2(t1.x + t3.x, t1.y + t3.y);
      out[((j0 + 2 * ns) << lj) + bb] = make_float2(t0.x - t2.x, t0.y - t2.y);
      out[((j0 + 3 * ns) << lj) + bb] = make_float2(t1.x - t3.x, t1.y - t3.y);
    }
    __syncthreads();
    float2* t = in;
    in = out;
    out = t;
  }
  if (s < log2n) {
    const int h = n >> 1;
    for (int jj = threadIdx.x; jj < (h << lj); jj += kBlock) {
      const int j = jj >> lj, bb = jj & bmask;
      float2 w = tw[j & (q - 1)];
      if (j >= q) w = make_float2(-w.y, w.x);
      const float2 v0 = in[(j << lj) + bb];
      const float2 v1 = cmul(in[((j + h) << lj) + bb], w);
      out[(j << lj) + bb] = make_float2(v0.x + v1.x, v0.y + v1.y);
      out[((j + h) << lj) + bb] = make_float2(v0.x - v1.x, v0.y - v1.y);
    }
    __syncthreads();
    float2* t = in;
    in = out;
    out = t;
  }
  return in;
}

__device__ __forceinline__ void fill_twiddles(float2* tw, int n) {
  for (int k = threadIdx.x; k < n / 4; k += kBlock) {
    float s, c;
    sincospif(2.0f * (float)k / (float)n, &s, &c);
    tw[k] = make_float2(c, s);
  }
}

}  // namespace mrx_dev
