// Device code shared by the spectral generators (screens, detector noise):
// Philox-4x32-10, Box-Muller, and the in-LDS Stockham inverse FFT.
#pragma once

#include "mrx_internal.h"

namespace mrx_dev {

using ::mrx_dev_common::kBlock;

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
  // v_sqrt_f32 as it is (1 ulp): sqrtf() adds a dozen instructions of denormal scaling and rounding repair
  const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln u1
  return make_float2(rad * __builtin_amdgcn_cosf(u2), rad * __builtin_amdgcn_sinf(u2));
}

// ---- in-LDS Stockham inverse FFT, radix 8 (+ one radix-4 or radix-2 pass) ---
// Autosort (natural order in and out, no bit reversal), ping-pong between two
// LDS images of n complex values.  tw[m] = exp(+2 pi i m / n) for m < n/4; a
// butterfly looks up w = exp(2 pi i k / (R ns)) and forms its powers by
// multiplication.  Unnormalised (numpy.fft.ifft * n).  Returns the image that
// holds the result.
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// a b + c in four fused multiply-adds (cadd(c, cmul(a, b)) takes six instructions: nothing may be reassociated)
__device__ __forceinline__ float2 cfma(float2 a, float2 b, float2 c) {
  return make_float2(__builtin_fmaf(-a.y, b.y, __builtin_fmaf(a.x, b.x, c.x)), __builtin_fmaf(a.y, b.x, __builtin_fmaf(a.x, b.y, c.y)));
}

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul_i(float2 a) { return make_float2(-a.y, a.x); }  // i a

// 8-point inverse DFT in place: v[m] <- sum_r v[r] exp(2 pi i r m / 8)
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
  constexpr float h = 0.70710678118654752f;
  // radix-2 layers on (r = r0 + 2 r1 + 4 r2): first over r2, then r1, then r0
  const float2 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
  const float2 a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
  const float2 a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
  const float2 a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
  // even outputs from (a0..a3): 4-point DFT; odd outputs from (a4..a7) with exp(2 pi i r / 8)
  const float2 b0 = cadd(a0, a2), b2 = csub(a0, a2);
  const float2 b1 = cadd(a1, a3), b3 = cmul_i(csub(a1, a3));
  const float2 c5 = make_float2(h * (a5.x - a5.y), h * (a5.x + a5.y));    // a5 e^{i pi/4}
  const float2 c6 = cmul_i(a6);                                            // a6 e^{i pi/2}
  const float2 c7 = make_float2(-h * (a7.x + a7.y), h * (a7.x - a7.y));   // a7 e^{3 i pi/4}
  const float2 d4 = cadd(a4, c6), d6 = csub(a4, c6);
  const float2 d5 = cadd(c5, c7), d7 = cmul_i(csub(c5, c7));
  v[0] = cadd(b0, b1);
  v[4] = csub(b0, b1);
  v[2] = cadd(b2, b3);
  v[6] = csub(b2, b3);
  v[1] = cadd(d4, d5);
  v[5] = csub(d4, d5);
  v[3] = cadd(d6, d7);
  v[7] = csub(d6, d7);
}

// twiddle exp(2 pi i idx / n) for 0 <= idx < n from the quarter table (q = n/4 entries)
__device__ __forceinline__ float2 tw_at(const float2* tw, int idx, int q) {
  const float2 w = tw[idx & (q - 1)];
  const int quad = idx / q;
  return quad == 0 ? w : quad == 1 ? make_float2(-w.y, w.x) : quad == 2 ? make_float2(-w.x, -w.y) : make_float2(w.y, -w.x);
}

// One Stockham pass of radix R in {2, 4, 8} over 2^lj interleaved sequences of length n:
// butterfly j (< n/R) of sequence b reads in[(j + r n/R) << lj | b], multiplies by
// exp(2 pi i r k / (R ns)), k = j mod ns, and writes out[((j - k) R + k + m ns) << lj | b].
template <int R, int kN = 0, int kThreads = kBlock>
__device__ __forceinline__ void stockham_pass(const float2* in, float2* out, const float2* tw, int n_rt,
                                              int ns, int lj) {
  // kN > 0: the length is a compile-time constant (and lj = 0), so the butterfly loop has a
  // fixed trip count and unrolls: all of a thread's LDS reads are in flight together, which
  // is what a kernel at two waves per SIMD needs
  const int n = kN > 0 ? kN : n_rt;
  const int per = n / R;          // butterflies per sequence
  const int q = n >> 2;
  const int bmask = (1 << lj) - 1;
  const int tstride = per / ns;   // n / (R ns): table index of exp(2 pi i / (R ns))
  auto butterfly = [&](int jj) {
    const int j = jj >> lj, b = jj & bmask;
    const int k = j & (ns - 1);
    float2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = in[((j + r * per) << lj) + b];
    if (ns > 1) {
      if constexpr (R == 2) {
        v[1] = cmul(v[1], tw_at(tw, k * tstride, q));
      } else {
        // k tstride < n / R: first quadrant for R >= 4
        const float2 w1 = tw[k * tstride];
        const float2 w2 = cmul(w1, w1);
        const float2 w3 = cmul(w2, w1);
        v[1] = cmul(v[1], w1);
        v[2] = cmul(v[2], w2);
        v[3] = cmul(v[3], w3);
        if constexpr (R == 8) {
          const float2 w4 = cmul(w2, w2);
          v[4] = cmul(v[4], w4);
          v[5] = cmul(v[5], cmul(w4, w1));
          v[6] = cmul(v[6], cmul(w4, w2));
          v[7] = cmul(v[7], cmul(w4, w3));
        }
      }
    }
    if constexpr (R == 8) {
      dft8(v);
    } else if constexpr (R == 4) {
      const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
      const float2 t2 = cadd(v[1], v[3]), t3 = cmul_i(csub(v[1], v[3]));
      v[0] = cadd(t0, t2);
      v[1] = cadd(t1, t3);
      v[2] = csub(t0, t2);
      v[3] = csub(t1, t3);
    } else {
      const float2 t0 = cadd(v[0], v[1]), t1 = csub(v[0], v[1]);
      v[0] = t0;
      v[1] = t1;
    }
    const int j0 = (j - k) * R + k;
#pragma unroll
    for (int m = 0; m < R; ++m) out[((j0 + m * ns) << lj) + b] = v[m];
  };
  if constexpr (kN >= R * kThreads) {
    constexpr int kTrips = kN / R / kThreads;
#pragma unroll
    for (int it = 0; it < kTrips; ++it) butterfly(threadIdx.x + it * kThreads);
  } else {
    for (int jj = threadIdx.x; jj < (per << lj); jj += kThreads) butterfly(jj);
  }
  __syncthreads();
}

// Passes of radix kRadix (8 or 4) while enough bits remain, then one smaller pass.  Which
// radix is faster depends on the caller's mix of VALU and LDS work: measured, radix 8 for the
// screens (n up to 8192, little else in the kernel) and radix 4 for the noise spectra.
template <int kRadix = 8, int kN = 0, int kThreads = kBlock>
__device__ __forceinline__ float2* fft_lds_inverse_batched(float2* a, float2* b,
                                                           const float2* tw, int n,
                                                           int log2n, int lj) {
  constexpr int kBits = kRadix == 8 ? 3 : 2;
  float2* in = a;
  float2* out = b;
  int ns = 1, s = 0;
  for (; s + kBits <= log2n; s += kBits, ns <<= kBits) {
    stockham_pass<kRadix, kN, kThreads>(in, out, tw, n, ns, lj);
    float2* t = in;
    in = out;
    out = t;
  }
  if (log2n - s == 2) {
    stockham_pass<4, kN, kThreads>(in, out, tw, n, ns, lj);
    return out;
  }
  if (log2n - s == 1) {
    stockham_pass<2, kN, kThreads>(in, out, tw, n, ns, lj);
    return out;
  }
  return in;
}

template <int kRadix = 8, int kN = 0, int kThreads = kBlock>
__device__ __forceinline__ float2* fft_lds_inverse(float2* a, float2* b, const float2* tw, int n,
                                                   int log2n) {
  return fft_lds_inverse_batched<kRadix, kN, kThreads>(a, b, tw, n, log2n, 0);
}

// ---- 64-point inverse FFT in registers --------------------------------------
// Radix-4 decimation in frequency, fully unrolled: every index and twiddle is a
// compile-time constant, so the arrays live in VGPRs and the trivial twiddles
// cost nothing.  Output k ends up at position rev4(k) (its three base-4 digits
// reversed): read x[k] as re[fft64_pos(k)].  Unnormalised, sign +.
constexpr float kCos64[64] = {1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f, 6.123233996e-17f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f, -1.000000000e+00f, -9.951847267e-01f, -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f, -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f, -1.950903220e-01f, -9.801714033e-02f, -1.836970199e-16f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f, 7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f};
constexpr float kSin64[64] = {0.000000000e+00f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f, 7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f, 1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f, 1.224646799e-16f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f, -1.000000000e+00f, -9.951847267e-01f, -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f, -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f, -1.950903220e-01f, -9.801714033e-02f};

__host__ __device__ constexpr int fft64_pos(int k) {
  return ((k & 3) << 4) | (k & 12) | ((k >> 4) & 3);
}

template <int L>
__device__ __forceinline__ void fft64_stage(float (&re)[64], float (&im)[64]) {
  constexpr int Q = L / 4;
#pragma unroll
  for (int s = 0; s < 64; s += L) {
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const int i0 = s + j, i1 = i0 + Q, i2 = i0 + 2 * Q, i3 = i0 + 3 * Q;
      const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
      const float cr = re[i2], ci = im[i2], dr = re[i3], di = im[i3];
      const float t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
      const float t2r = br + dr, t2i = bi + di;
      const float t3r = -(bi - di), t3i = br - dr;  // +i (b - d)
      re[i0] = t0r + t2r;
      im[i0] = t0i + t2i;
      const float y1r = t1r + t3r, y1i = t1i + t3i;
      const float y2r = t0r - t2r, y2i = t0i - t2i;
      const float y3r = t1r - t3r, y3i = t1i - t3i;
      if (j == 0) {
        re[i1] = y1r; im[i1] = y1i; re[i2] = y2r; im[i2] = y2i; re[i3] = y3r; im[i3] = y3i;
      } else {
        constexpr int step = 64 / L;  // twiddle exp(2 pi i j m / L) = table[j m step]
        const float c1 = kCos64[(j * step) & 63], s1 = kSin64[(j * step) & 63];
        const float c2 = kCos64[(2 * j * step) & 63], s2 = kSin64[(2 * j * step) & 63];
        const float c3 = kCos64[(3 * j * step) & 63], s3 = kSin64[(3 * j * step) & 63];
        re[i1] = y1r * c1 - y1i * s1; im[i1] = y1r * s1 + y1i * c1;
        re[i2] = y2r * c2 - y2i * s2; im[i2] = y2r * s2 + y2i * c2;
        re[i3] = y3r * c3 - y3i * s3; im[i3] = y3r * s3 + y3i * c3;
      }
    }
  }
}

__device__ __forceinline__ void fft64_inverse_reg(float (&re)[64], float (&im)[64]) {
  fft64_stage<64>(re, im);
  fft64_stage<16>(re, im);
  fft64_stage<4>(re, im);
}

// ---- 4096-point inverse FFT of a 256-thread workgroup, 16 values per thread in registers ----
// Three radix-16 register transforms and two exchanges through LDS (4 LDS accesses per value and
// two barriers, against 12 accesses and six barriers of the radix-4 Stockham passes above).
constexpr float kCos16[16] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
                              -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f, 0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
constexpr float kSin16[16] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                              0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};

// a exp(2 pi i P / 16), P a compile-time constant: the multiples of a quarter turn are moves,
// the odd multiples of an eighth two additions and two multiplications
template <int P>
__device__ __forceinline__ float2 mul_w16(float2 a) {
  constexpr int p = P & 15;
  constexpr float h = 0.70710678118654752f;
  if constexpr (p == 0) return a;
  else if constexpr (p == 4) return make_float2(-a.y, a.x);
  else if constexpr (p == 8) return make_float2(-a.x, -a.y);
  else if constexpr (p == 12) return make_float2(a.y, -a.x);
  else if constexpr (p == 2) return make_float2(h * (a.x - a.y), h * (a.x + a.y));
  else if constexpr (p == 6) return make_float2(-h * (a.x + a.y), h * (a.x - a.y));
  else if constexpr (p == 10) return make_float2(h * (a.y - a.x), -h * (a.x + a.y));
  else if constexpr (p == 14) return make_float2(h * (a.x + a.y), h * (a.y - a.x));
  else return make_float2(a.x * kCos16[p] - a.y * kSin16[p], a.x * kSin16[p] + a.y * kCos16[p]);
}

// y[m] = sum_r x[r] i^(r m), in place
__device__ __forceinline__ void radix4_inverse(float2& x0, float2& x1, float2& x2, float2& x3) {
  const float2 t0 = cadd(x0, x2), t1 = csub(x0, x2);
  const float2 t2 = cadd(x1, x3), t3 = cmul_i(csub(x1, x3));
  x0 = cadd(t0, t2);
  x1 = cadd(t1, t3);
  x2 = csub(t0, t2);
  x3 = csub(t1, t3);
}

// 16-point inverse DFT in place: output d ends up at position dft16_pos(d) (its two base-4
// digits swapped).  b = 4 b1 + b0, d = 4 d1 + d0: w^(b d) = i^(b1 d0) i^(b0 d1) w^(b0 d0).
__host__ __device__ constexpr int dft16_pos(int d) { return ((d & 3) << 2) | (d >> 2); }

__device__ __forceinline__ void dft16(float2 (&v)[16]) {
  // over b1 for every b0: v[b0 + 4 d0] <- sum_b1 v[b0 + 4 b1] i^(b1 d0)
  radix4_inverse(v[0], v[4], v[8], v[12]);
  radix4_inverse(v[1], v[5], v[9], v[13]);
  radix4_inverse(v[2], v[6], v[10], v[14]);
  radix4_inverse(v[3], v[7], v[11], v[15]);
  // w^(b0 d0)
  v[5] = mul_w16<1>(v[5]);
  v[6] = mul_w16<2>(v[6]);
  v[7] = mul_w16<3>(v[7]);
  v[9] = mul_w16<2>(v[9]);
  v[10] = mul_w16<4>(v[10]);
  v[11] = mul_w16<6>(v[11]);
  v[13] = mul_w16<3>(v[13]);
  v[14] = mul_w16<6>(v[14]);
  v[15] = mul_w16<9>(v[15]);
  // over b0 for every d0: v[d1 + 4 d0] <- sum_b0 v[b0 + 4 d0] i^(b0 d1)
  radix4_inverse(v[0], v[1], v[2], v[3]);
  radix4_inverse(v[4], v[5], v[6], v[7]);
  radix4_inverse(v[8], v[9], v[10], v[11]);
  radix4_inverse(v[12], v[13], v[14], v[15]);
}

__device__ __forceinline__ float2 csq(float2 a) { return make_float2(a.x * a.x - a.y * a.y, 2.0f * a.x * a.y); }

// w[d] = w1^d, d < 16, by squarings and products four deep (error ~4 ulp)
__device__ __forceinline__ void powers16(float2 w1, float2 (&w)[16]) {
  w[0] = make_float2(1.0f, 0.0f);
  w[1] = w1;
  w[2] = csq(w1);
  w[3] = cmul(w[2], w1);
  w[4] = csq(w[2]);
  w[5] = cmul(w[4], w1);
  w[6] = csq(w[3]);
  w[7] = cmul(w[4], w[3]);
  w[8] = csq(w[4]);
  w[9] = cmul(w[8], w1);
  w[10] = csq(w[5]);
  w[11] = cmul(w[8], w[3]);
  w[12] = csq(w[6]);
  w[13] = cmul(w[8], w[5]);
  w[14] = csq(w[7]);
  w[15] = cmul(w[8], w[7]);
}

constexpr int kFft4096Pitch = 17;                                 // float2 per 16 values: both exchanges conflict-free
constexpr int kFft4096Image = 4096 / 16 * kFft4096Pitch;          // float2 per exchange image (34 816 bytes)

// in: v[b] = x[t + 256 b]; out: v[dft16_pos(f)] = y[t + 256 f], y[j] = sum_k x[k] exp(2 pi i j k / 4096).
// k = t + 256 b, j = d + 16 c:  exp(2 pi i k j / 4096) = exp(2 pi i t d / 4096) exp(2 pi i t c / 256) exp(2 pi i b d / 16);
// t = u + 16 v, c = e + 16 f:  exp(2 pi i t c / 256)  = exp(2 pi i u e / 256)  exp(2 pi i u f / 16)  exp(2 pi i v e / 16).
// ex1 and ex2: kFft4096Image float2 each.  A caller that loops needs no barrier of its own: the
// next call writes ex1 after this call's second barrier, which every read of ex1 precedes, and
// ex2 after its own first barrier, which every thread reaches after its reads of ex2 here.
__device__ __forceinline__ void fft4096_workgroup(float2 (&v)[16], float2* ex1, float2* ex2) {
  const int t = threadIdx.x, lo = t & 15, hi = t >> 4;
  float2 w[16];
  dft16(v);  // over b
  powers16(make_float2(__builtin_amdgcn_cosf((float)t * (1.0f / 4096.0f)), __builtin_amdgcn_sinf((float)t * (1.0f / 4096.0f))), w);
#pragma unroll
  for (int d = 0; d < 16; ++d) ex1[t * kFft4096Pitch + d] = d ? cmul(v[dft16_pos(d)], w[d]) : v[0];
  __syncthreads();
  // u = lo, d = hi: over v
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = ex1[(lo + 16 * q) * kFft4096Pitch + hi];
  dft16(v);
  powers16(make_float2(__builtin_amdgcn_cosf((float)lo * (1.0f / 256.0f)), __builtin_amdgcn_sinf((float)lo * (1.0f / 256.0f))), w);
#pragma unroll
  for (int e = 0; e < 16; ++e) ex2[(hi + 16 * e) * kFft4096Pitch + lo] = e ? cmul(v[dft16_pos(e)], w[e]) : v[0];
  __syncthreads();
  // d = lo, e = hi (d + 16 e = t): over u
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = ex2[t * kFft4096Pitch + q];
  dft16(v);
}

// ---- the same scheme for n = 256 RB points, RB = 4, 8, 16, by T = 16 RB threads -------------
// (RB = 16 is fft4096_workgroup).  k = t + T b, j = d + 16 c:
//   exp(2 pi i k j / n) = exp(2 pi i t d / n) exp(2 pi i t c / T) exp(2 pi i b d / 16);
// t = u + 16 v (v < RB), c = e + RB f (e < RB, f < 16):
//   exp(2 pi i t c / T) = exp(2 pi i u e / T) exp(2 pi i u f / 16) exp(2 pi i v e / RB).
// Pass A: 16-point transform over b and the twiddle; pass B: 16/RB transforms of RB points over v
// per thread (thread (u, dg) takes d = dg + RB g) and the twiddle; pass C: 16-point transform over
// u.  in: v[b] = x[t + T b]; out: v[dft16_pos(f)] = y[t + T f].  ex1, ex2: 17 T float2 each, the
// transform's own; every thread of the workgroup calls it (two workgroup barriers inside), so a
// workgroup of 256 threads runs 16 / RB transforms side by side.
// kOneImage: ex2 is not used -- the second exchange goes through ex1 again, behind a barrier of its own (every thread
// reads its 16 values of the first exchange into registers, all meet, then write), and a last barrier lets the caller
// write ex1 at once.  Half the LDS (17 T float2 a transform: 34.8 KB a workgroup instead of 69.6: four workgroups a CU
// instead of two) for two more barriers: for callers whose occupancy the images bound (mrx_screen.hip).
template <int RB, bool kOneImage = false>
__device__ __forceinline__ void fft_regs(float2 (&v)[16], float2* ex1, float2* ex2, int t) {
  static_assert(RB == 4 || RB == 8 || RB == 16, "256 RB points, RB = 4, 8 or 16");
  constexpr int T = 16 * RB, G = 16 / RB;
  constexpr float inv_n = 1.0f / (float)(256 * RB), inv_t = 1.0f / (float)T;
  const int lo = t & 15, hi = t >> 4;
  float2 w[16];
  dft16(v);
  powers16(make_float2(__builtin_amdgcn_cosf((float)t * inv_n), __builtin_amdgcn_sinf((float)t * inv_n)), w);
#pragma unroll
  for (int d = 0; d < 16; ++d) ex1[t * kFft4096Pitch + d] = d ? cmul(v[dft16_pos(d)], w[d]) : v[0];
  __syncthreads();
  powers16(make_float2(__builtin_amdgcn_cosf((float)lo * inv_t), __builtin_amdgcn_sinf((float)lo * inv_t)), w);
  float2* const second = kOneImage ? ex1 : ex2;
  float2 a[G][RB];
  if constexpr (kOneImage) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int q = 0; q < RB; ++q) a[g][q] = ex1[(lo + 16 * q) * kFft4096Pitch + hi + RB * g];
    __syncthreads();  // every read of the first exchange precedes the writes of the second
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int d = hi + RB * g;
    if constexpr (!kOneImage) {
#pragma unroll
      for (int q = 0; q < RB; ++q) a[g][q] = ex1[(lo + 16 * q) * kFft4096Pitch + d];
    }
    if constexpr (RB == 16) {
      dft16(a[g]);
#pragma unroll
      for (int e = 0; e < RB; ++e) second[(d + 16 * e) * kFft4096Pitch + lo] = e ? cmul(a[g][dft16_pos(e)], w[e]) : a[g][0];
    } else {
      if constexpr (RB == 8) dft8(a[g]);
      else radix4_inverse(a[g][0], a[g][1], a[g][2], a[g][3]);
#pragma unroll
      for (int e = 0; e < RB; ++e) second[(d + 16 * e) * kFft4096Pitch + lo] = e ? cmul(a[g][e], w[e]) : a[g][0];
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = second[t * kFft4096Pitch + q];
  if constexpr (kOneImage) __syncthreads();  // the image is the caller's again
  dft16(v);
}

template <int kThreads = kBlock>
__device__ __forceinline__ void fill_twiddles(float2* tw, int n) {
  for (int k = threadIdx.x; k < n / 4; k += kThreads) {
    float s, c;
    sincospif(2.0f * (float)k / (float)n, &s, &c);
    tw[k] = make_float2(c, s);
  }
}

}  // namespace mrx_dev
