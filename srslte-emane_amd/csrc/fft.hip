// Batched mixed-radix FFT + OFDM modulator/demodulator + SC-FDMA transform precoding for gfx950.
//
// Replaces fftwf_execute on the guru plans of ofdm.c:90-101 and the per-symbol copies/normalisation of
// srslte_ofdm_rx_slot / srslte_ofdm_tx_slot (ofdm.c:398-422, :488-530), the half-carrier shift
// (ofdm.c:360-378, :455-457, :591-593) and srslte_dft_precoding (dft_precoding.c:100-113).
//
// Design: one workgroup per transform. The N-point sequence lives in LDS (two ping-pong buffers, Stockham
// autosort, radices 4/2/3/5 so 128..2048, 3*2^n and every 12*2^a3^b5^c size is covered). CP strip, DC skip,
// guard strip, fftshift and 1/sqrt(N) are fused into the global load/store indexing, so HBM traffic is exactly
// the algorithmic bytes: each time sample is read once, each used bin written once, both coalesced.
#include "common.hpp"
#include "phy_hip_internal.hpp"
#include <math.h>
#include <map>
#include <mutex>
#include <vector>

namespace {

__device__ __forceinline__ cf32 cmul(cf32 a, cf32 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cf32 cadd(cf32 a, cf32 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf32 csub(cf32 a, cf32 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * (j*s)
__device__ __forceinline__ cf32 cmulj(cf32 a, float s) { return make_float2(-s * a.y, s * a.x); }
// a * (c + j*sgn*s): multiplication by exp(sgn * j * theta) with cos/sin given
__device__ __forceinline__ cf32 cmulw(cf32 a, float c, float s, float sgn) { return make_float2(a.x * c - sgn * s * a.y, a.y * c + sgn * s * a.x); }

// Natural-order R-point DFT in registers: v[r] <- sum_q v[q] exp(sgn*j*2*pi*q*r/R).
template <int R>
__device__ __forceinline__ void dft_r(cf32* v, float sgn)
{
  if constexpr (R == 2) {
    cf32 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  } else if constexpr (R == 4) {
    cf32 a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = cmulj(csub(v[1], v[3]), sgn);
    v[0] = cadd(a, c);
    v[1] = cadd(b, d);
    v[2] = csub(a, c);
    v[3] = csub(b, d);
  } else if constexpr (R == 3) {
    const float s = 0.86602540378443864676f;
    cf32 t = cadd(v[1], v[2]), u = cmulj(csub(v[1], v[2]), sgn * s);
    cf32 m = make_float2(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
    v[0]   = cadd(v[0], t);
    v[1]   = cadd(m, u);
    v[2]   = csub(m, u);
  } else if constexpr (R == 5) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f, s1 = 0.95105651629515357212f,
                s2 = 0.58778525229247312917f;
    cf32 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), d1 = csub(v[1], v[4]), d2 = csub(v[2], v[3]);
    cf32 m1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
    cf32 m2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
    cf32 u1 = cmulj(make_float2(s1 * d1.x + s2 * d2.x, s1 * d1.y + s2 * d2.y), sgn);
    cf32 u2 = cmulj(make_float2(s2 * d1.x - s1 * d2.x, s2 * d1.y - s1 * d2.y), sgn);
    v[0]    = cadd(v[0], cadd(t1, t2));
    v[1]    = cadd(m1, u1);
    v[4]    = csub(m1, u1);
    v[2]    = cadd(m2, u2);
    v[3]    = csub(m2, u2);
  } else if constexpr (R == 6) { // 6 = 2 x 3: n = 3 n1 + n2, k = k1 + 2 k2
    cf32 a[3], b[3];           // a: k1 = 0, b: k1 = 1 (index n2)
#pragma unroll
    for (int n2 = 0; n2 < 3; n2++) {
      a[n2] = cadd(v[n2], v[3 + n2]);
      b[n2] = csub(v[n2], v[3 + n2]);
    }
    b[1] = cmulw(b[1], 0.5f, 0.86602540378443864676f, sgn);  // W6^1
    b[2] = cmulw(b[2], -0.5f, 0.86602540378443864676f, sgn); // W6^2
    dft_r<3>(a, sgn);
    dft_r<3>(b, sgn);
#pragma unroll
    for (int k2 = 0; k2 < 3; k2++) {
      v[2 * k2]     = a[k2];
      v[2 * k2 + 1] = b[k2];
    }
  } else if constexpr (R == 8) { // even/odd split
    cf32 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    dft_r<4>(e, sgn);
    dft_r<4>(o, sgn);
    const float h = 0.70710678118654752440f;
    o[1] = cmulw(o[1], h, h, sgn);
    o[2] = cmulj(o[2], sgn);
    o[3] = cmulw(o[3], -h, h, sgn);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      v[k]     = cadd(e[k], o[k]);
      v[k + 4] = csub(e[k], o[k]);
    }
  } else { // 16 = 4 x 4: n = 4 n1 + n2, k = k1 + 4 k2
    static_assert(R == 16, "unsupported radix");
    cf32 a[4][4]; // a[n2][k1]
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) {
      cf32 t[4] = {v[n2], v[4 + n2], v[8 + n2], v[12 + n2]};
      dft_r<4>(t, sgn);
#pragma unroll
      for (int k1 = 0; k1 < 4; k1++) a[n2][k1] = t[k1];
    }
    // W16^(n2*k1): cos/sin of 2*pi*m/16
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    a[1][1] = cmulw(a[1][1], c1, s1, sgn);   // m=1
    a[1][2] = cmulw(a[1][2], h, h, sgn);     // m=2
    a[1][3] = cmulw(a[1][3], s1, c1, sgn);   // m=3
    a[2][1] = cmulw(a[2][1], h, h, sgn);     // m=2
    a[2][2] = cmulj(a[2][2], sgn);           // m=4
    a[2][3] = cmulw(a[2][3], -h, h, sgn);    // m=6
    a[3][1] = cmulw(a[3][1], s1, c1, sgn);   // m=3
    a[3][2] = cmulw(a[3][2], -h, h, sgn);    // m=6
    a[3][3] = cmulw(a[3][3], -c1, -s1, sgn); // m=9: cos = -c1, sin = -s1
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) {
      cf32 t[4] = {a[0][k1], a[1][k1], a[2][k1], a[3][k1]};
      dft_r<4>(t, sgn);
#pragma unroll
      for (int k2 = 0; k2 < 4; k2++) v[k1 + 4 * k2] = t[k2];
    }
  }
}

// LDS index with one pad element per 16: the radix-16 pass writes with a stride of 16 elements across lanes, which would
// put a whole wave on two banks; i + i/16 spreads it over all of them (cdna_hip_programming.md Guideline 4).
__device__ __forceinline__ int pad16(int i) { return i + (i >> 4); }

// One Stockham autosort pass of radix R: butterfly j reads x[j + r*N/R], multiplies by w^(r*k), k = j mod Ns, and writes
// y[(j/Ns)*Ns*R + k + r*Ns]. Ld/St abstract where x and y live: the FIRST pass reads the caller's global input and the
// LAST writes the caller's global output (with whatever index mapping the caller fuses in), the others use LDS.
template <int R, bool MID_INPLACE, typename Ld, typename St>
__device__ __forceinline__ void stockham_pass(int N, int Ns, const cf32* __restrict__ tw, float sgn, Ld ld, St st)
{
  const int nb = N / R, tstep = N / (Ns * R);
  auto      body = [&](int j, cf32* v) {
    const int k = j % Ns;
    if (k != 0) {
      cf32 w1 = tw[k * tstep]; // exp(-j*2*pi*k/(Ns*R)); powers by repeated multiplication (<= 15 products, ~1e-6 relative)
      w1.y    = -sgn * w1.y;
      cf32 w  = w1;
#pragma unroll
      for (int r = 1; r < R; r++) {
        v[r] = cmul(v[r], w);
        if (r + 1 < R) w = cmul(w, w1);
      }
    }
    dft_r<R>(v, sgn);
    const int j0 = (j / Ns) * Ns * R + k;
#pragma unroll
    for (int r = 0; r < R; r++) st(j0 + r * Ns, v[r]);
  };
  if constexpr (MID_INPLACE) { // LDS -> same LDS buffer, at most one butterfly per thread: everyone loads, barrier, everyone stores
    const int  j      = threadIdx.x;
    const bool active = j < nb;
    cf32       v[R];
    if (active) {
#pragma unroll
      for (int r = 0; r < R; r++) v[r] = ld(j + r * nb);
    }
    __syncthreads();
    if (active) body(j, v);
  } else {
    for (int j = threadIdx.x; j < nb; j += blockDim.x) {
      cf32 v[R];
#pragma unroll
      for (int r = 0; r < R; r++) v[r] = ld(j + r * nb);
      body(j, v);
    }
  }
}

// Generic plan (any N = 2^a 3^b 5^c, run-time radix list, radices 2..5 only to keep the code small): two LDS buffers.
// gld(n): element n of the input sequence; gst(n, v): element n of the transform.
template <typename GLd, typename GSt>
__device__ __forceinline__ void fft_passes(const FftFactors& f, const cf32* __restrict__ tw, float sgn, cf32* lds, GLd gld, GSt gst)
{
  const int N = f.N, Np = pad16(N) + 1;
  int       Ns = 1;
  for (int i = 0; i < f.nf; i++) {
    const bool first = i == 0, last = i == f.nf - 1;
    cf32*      src   = lds + ((i + 1) & 1) * Np;
    cf32*      dst   = lds + (i & 1) * Np;
    auto       ld    = [&](int n) { return first ? gld(n) : src[pad16(n)]; };
    auto       st    = [&](int n, cf32 v) {
      if (last) {
        gst(n, v);
      } else {
        dst[pad16(n)] = v;
      }
    };
    switch (f.radix[i]) {
      case 5: stockham_pass<5, false>(N, Ns, tw, sgn, ld, st); break;
      case 4: stockham_pass<4, false>(N, Ns, tw, sgn, ld, st); break;
      case 3: stockham_pass<3, false>(N, Ns, tw, sgn, ld, st); break;
      default: stockham_pass<2, false>(N, Ns, tw, sgn, ld, st); break;
    }
    Ns *= f.radix[i];
    if (!last) __syncthreads();
  }
}

// Fixed plan N = R0*R1*R2 (R2 = 1: two passes) for the OFDM sizes: global -> registers -> LDS, LDS -> LDS in place
// (one radix-R1 butterfly per thread), LDS -> registers -> global. One LDS buffer, two barriers-and-a-half, no run-time
// dispatch: everything about the plan is a compile-time constant.
template <int R0, int R1, int R2, typename GLd, typename GSt>
__device__ __forceinline__ void fft_fixed(const cf32* __restrict__ tw, float sgn, cf32* lds, GLd gld, GSt gst)
{
  constexpr int N = R0 * R1 * R2;
  auto lld = [&](int n) { return lds[pad16(n)]; };
  auto lst = [&](int n, cf32 v) { lds[pad16(n)] = v; };
  stockham_pass<R0, false>(N, 1, tw, sgn, gld, lst);
  __syncthreads();
  if constexpr (R2 == 1) {
    stockham_pass<R1, false>(N, R0, tw, sgn, lld, gst);
  } else {
    stockham_pass<R1, true>(N, R0, tw, sgn, lld, lst);
    __syncthreads();
    stockham_pass<R2, false>(N, R0 * R1, tw, sgn, lld, gst);
  }
}

// Runs the fixed plan when the launch was made for one (R0 != 0), the generic plan otherwise.
template <int R0, int R1, int R2, typename GLd, typename GSt>
__device__ __forceinline__ void fft_any(const FftFactors& f, const cf32* __restrict__ tw, float sgn, cf32* lds, GLd gld, GSt gst)
{
  if constexpr (R0 != 0) {
    fft_fixed<R0, R1, R2>(tw, sgn, lds, gld, gst);
  } else {
    fft_passes(f, tw, sgn, lds, gld, gst);
  }
}

struct OfdmGeom {
  FftFactors f;
  int        nof_re, nsym, sf_len, dc, cp_max, sym0; // sym0: first symbol of this launch (slot calls)
  float      norm; // 1 or 1/sqrt(N)
  int        sym_off[14]; // first sample (start of CP) of each symbol in the subframe
  int        cp_len[14];
};

constexpr int FFT_THREADS = 128;
#define FFT_BOUNDS __launch_bounds__(FFT_THREADS)

// grid = (nsym, nof_sf). in: [nof_sf][sf_len] time samples; out: [nof_sf][nsym][nof_re] resource grid.
// CP strip and the half-carrier shift ride on the first pass's global loads; guard/DC strip, half swap and 1/sqrt(N) on
// the last pass's global stores (ofdm.c:410-416): every sample is read once and every used bin written once.
template <int R0, int R1, int R2>
__global__ FFT_BOUNDS void ofdm_rx_kernel(const cf32* __restrict__ in, cf32* __restrict__ out, OfdmGeom g,
                                                              const cf32* __restrict__ tw, const cf32* __restrict__ shift)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int   s = blockIdx.x + g.sym0, sf = blockIdx.y, N = g.f.N, half = g.nof_re / 2;
  const cf32* src = in + (size_t)sf * g.sf_len + g.sym_off[s] + g.cp_len[s];
  cf32*       dst = out + ((size_t)sf * g.nsym + s) * g.nof_re;
  fft_any<R0, R1, R2>(
      g.f, tw, -1.0f, reinterpret_cast<cf32*>(lds_raw),
      [&](int n) {
        cf32 v = src[n];
        if (shift) v = cmul(v, shift[g.cp_max + n]); // ofdm.c:369-371 with t - cplen = n
        return v;
      },
      [&](int n, cf32 v) { // ofdm.c:411-412
        v = make_float2(v.x * g.norm, v.y * g.norm);
        if (n >= N - half) {
          dst[n - (N - half)] = v;
        } else if (n >= g.dc && n < g.dc + half) {
          dst[half + n - g.dc] = v;
        }
      });
}

// in: [nof_sf][nsym][nof_re] grid; out: [nof_sf][sf_len] time samples with CP.
template <int R0, int R1, int R2>
__global__ FFT_BOUNDS void ofdm_tx_kernel(const cf32* __restrict__ in, cf32* __restrict__ out, OfdmGeom g,
                                                              const cf32* __restrict__ tw, const cf32* __restrict__ shift)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int   s = blockIdx.x + g.sym0, sf = blockIdx.y, N = g.f.N, half = g.nof_re / 2, cp = g.cp_len[s];
  const cf32* src = in + ((size_t)sf * g.nsym + s) * g.nof_re;
  cf32*       dst = out + (size_t)sf * g.sf_len + g.sym_off[s];
  fft_any<R0, R1, R2>(
      g.f, tw, 1.0f, reinterpret_cast<cf32*>(lds_raw),
      [&](int n) { // ofdm.c:509-515 (guards and DC stay zero)
        if (n >= g.dc && n < g.dc + half) return src[half + n - g.dc];
        if (n >= N - half) return src[n - (N - half)];
        return make_float2(0.f, 0.f);
      },
      [&](int n, cf32 v) { // ofdm.c:519-529: body, and the tail again as cyclic prefix
        v = make_float2(v.x * g.norm, v.y * g.norm);
        dst[cp + n] = shift ? cmul(v, shift[g.cp_max + n]) : v;
        if (n >= N - cp) {
          const int t = n - (N - cp);
          dst[t]      = shift ? cmul(v, shift[g.cp_max + t - cp]) : v;
        }
      });
}

// Generic batched c2c transform: howmany transforms, element strides 1, distances idist/odist, output * scale.
template <int R0, int R1, int R2>
__global__ FFT_BOUNDS void dft_batch_kernel(const cf32* __restrict__ in, cf32* __restrict__ out, FftFactors f,
                                                                int idist, int odist, float sgn, float scale,
                                                                const cf32* __restrict__ tw)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const cf32* src = in + (size_t)blockIdx.x * idist;
  cf32*       dst = out + (size_t)blockIdx.x * odist;
  fft_any<R0, R1, R2>(
      f, tw, sgn, reinterpret_cast<cf32*>(lds_raw), [&](int n) { return src[n]; },
      [&](int n, cf32 v) { dst[n] = make_float2(v.x * scale, v.y * scale); });
}

// Any other length (a prime factor beyond 5, or N > 2048): the DFT sum itself, one thread per output bin, double accumulation.
// FFTW plans every N (dft_fftw.c:93-117) and callers outside the hot path rely on it (PRACH: N_zc = 839 / 139, prach.c; the PSS / SSS
// correlators' conv_fft_cc lengths; utils/test/dft_test.c -N 255): they keep working through the boundary, at O(N^2) cost.
constexpr int DIRECT_CHUNK = 1024;
__global__ __launch_bounds__(256) void dft_direct_kernel(const cf32* __restrict__ in, cf32* __restrict__ out, int N, int idist, int odist, float sgn,
                                                         float scale, const cf32* __restrict__ tw)
{
  __shared__ cf32 xs[DIRECT_CHUNK];
  const cf32*     src = in + (size_t)blockIdx.y * idist;
  cf32*           dst = out + (size_t)blockIdx.y * odist;
  const int       k   = blockIdx.x * blockDim.x + threadIdx.x;
  double          ar = 0.0, ai = 0.0;
  int             idx = 0; // (n * k) mod N, kept incrementally
  const int       kk  = k < N ? k : 0;
  for (int n0 = 0; n0 < N; n0 += DIRECT_CHUNK) {
    const int len = min(DIRECT_CHUNK, N - n0);
    __syncthreads();
    for (int i = threadIdx.x; i < len; i += blockDim.x) xs[i] = src[n0 + i];
    __syncthreads();
    for (int i = 0; i < len; i++) {
      const cf32   w  = tw[idx];
      const double wr = w.x, wi = -sgn * w.y; // tw = exp(-j 2 pi k / N); sgn = -1 forward, +1 backward
      ar += (double)xs[i].x * wr - (double)xs[i].y * wi;
      ai += (double)xs[i].x * wi + (double)xs[i].y * wr;
      idx += kk;
      if (idx >= N) idx -= N;
    }
  }
  if (k < N) dst[k] = make_float2((float)(ar * scale), (float)(ai * scale));
}

// ---------------------------------------------------------------- host side: twiddle cache
struct TwEntry {
  FftFactors f;
  cf32*      d_tw;
};
std::mutex             g_tw_mutex;
std::map<long, TwEntry> g_tw_cache; // key: device * 2^20 + N
constexpr int           FFT_MAX_N = 1 << 17; // prach.c plans up to 12 x 2048 x 4 points

static int fft_threads(int N) { return N >= 1024 ? 128 : 64; }

// N with a compile-time plan (the OFDM symbol sizes of phy_common.c:304-345)
static bool fft_is_fixed(int N) { return N == 128 || N == 256 || N == 384 || N == 512 || N == 768 || N == 1024 || N == 1536 || N == 2048; }

int factorize(int N, FftFactors* f)
{ // generic plan: 4s, then 2, 3s, 5s
  f->N  = N;
  f->nf = 0;
  int n = N;
  while (n % 4 == 0 && f->nf < 8) { f->radix[f->nf++] = 4; n /= 4; }
  while (n % 2 == 0 && f->nf < 8) { f->radix[f->nf++] = 2; n /= 2; }
  while (n % 3 == 0 && f->nf < 8) { f->radix[f->nf++] = 3; n /= 3; }
  while (n % 5 == 0 && f->nf < 8) { f->radix[f->nf++] = 5; n /= 5; }
  f->inplace = fft_is_fixed(N) ? 1 : 0;
  if (n != 1 || N > 2048) f->nf = 0; // no LDS plan: dft_direct_kernel
  return 0;
}

static size_t fft_lds_bytes(const FftFactors& f) { return (f.inplace ? 1 : 2) * sizeof(cf32) * (size_t)(f.N + f.N / 16 + 2); }

// launches KERNEL<plan> for the transform size: fixed plans for the OFDM sizes, <0,0,0> = generic otherwise
#define FFT_DISPATCH(KERNEL, N, GRID, BLOCK, LDS, STREAM, ...)                                                        \
  switch (N) {                                                                                                        \
    case 128: hipLaunchKernelGGL((KERNEL<16, 8, 1>), GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                   \
    case 256: hipLaunchKernelGGL((KERNEL<16, 16, 1>), GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                  \
    case 384: hipLaunchKernelGGL((KERNEL<16, 8, 3>), GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                   \
    case 512: hipLaunchKernelGGL((KERNEL<16, 16, 2>), GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                  \
    case 768: hipLaunchKernelGGL((KERNEL<16, 16, 3>), GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                  \
    case 1024: hipLaunchKernelGGL((KERNEL<16, 16, 4>), GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                 \
    case 1536: hipLaunchKernelGGL((KERNEL<16, 16, 6>), GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                 \
    case 2048: hipLaunchKernelGGL((KERNEL<16, 16, 8>), GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                 \
    default: hipLaunchKernelGGL((KERNEL<0, 0, 0>), GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                     \
  }

} // namespace

int fft_get_plan(int N, FftFactors* f, const cf32** d_tw)
{
  if (N < 1 || N > FFT_MAX_N) {
    hip_log("[srslte_hip] unsupported DFT size %d (1..%d)\n", N, FFT_MAX_N);
    return SRSLTE_ERROR_INVALID_INPUTS;
  }
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_tw_mutex); // planning is serialised, like dft_fftw.c:42
  auto it = g_tw_cache.find(((long)dev << 20) + N);
  if (it == g_tw_cache.end()) {
    TwEntry e;
    factorize(N, &e.f);
    std::vector<cf32> tw(N);
    for (int k = 0; k < N; k++) tw[k] = make_float2((float)cos(2.0 * M_PI * k / N), (float)-sin(2.0 * M_PI * k / N));
    HIP_TRY(hipMalloc((void**)&e.d_tw, sizeof(cf32) * N));
    HIP_TRY(hipMemcpy(e.d_tw, tw.data(), sizeof(cf32) * N, hipMemcpyHostToDevice));
    it = g_tw_cache.emplace(((long)dev << 20) + N, e).first;
  }
  *f    = it->second.f;
  *d_tw = it->second.d_tw;
  return SRSLTE_SUCCESS;
}

// ---------------------------------------------------------------- OFDM object
struct srslte_hip_ofdm {
  OfdmGeom    g;  // regular subframe
  OfdmGeom    gm; // MBSFN subframe (slot 0: non-MBSFN region + guard + extended-CP symbols), valid when mbsfn
  bool        mbsfn;
  const cf32* d_tw;
  cf32*       d_shift; // [cp_max + N], exp(j 2 pi m f / N), m = -cp_max .. N-1; nullptr when no shift
  bool        is_rx, normalize;
  float       freq_shift_f;
  int         nof_prb;
};

extern "C" srslte_hip_ofdm_t* srslte_hip_ofdm_create(int nof_prb, int cp_is_norm, int is_rx)
{
  return srslte_hip_ofdm_create_sz(nof_prb, lte_symbol_sz(nof_prb), cp_is_norm, is_rx);
}

// The symbol size is the caller's to choose (srslte_ofdm_init_ takes it as an argument, ofdm.c:38-57): srslte_symbol_sz gives 128 / 256 / 384 /
// 768 / 1024 / 1536 by default and 128 / 256 / 512 / 1024 / 1536 / 2048 after srslte_use_standard_symbol_size(true) (phy_common.c:304-345, what
// rf_uhd_imp.c:457,:473 selects for some radios). Any of these sizes that holds the carriers: guards (N - 12 nof_prb) / 2, CP lengths scaled with N.
extern "C" srslte_hip_ofdm_t* srslte_hip_ofdm_create_sz(int nof_prb, int symbol_sz, int cp_is_norm, int is_rx)
{
  const int N = symbol_sz;
  const bool size_ok = N == 128 || N == 256 || N == 384 || N == 512 || N == 768 || N == 1024 || N == 1536 || N == 2048;
  if (nof_prb <= 0 || nof_prb > 110 || !size_ok || 12 * nof_prb >= N) {
    hip_log("[srslte_hip] Error: Invalid nof_prb=%d / symbol_sz=%d\n", nof_prb, symbol_sz);
    return nullptr;
  }
  auto* q = new srslte_hip_ofdm();
  if (fft_get_plan(N, &q->g.f, &q->d_tw)) {
    delete q;
    return nullptr;
  }
  const int nsym_slot = cp_is_norm ? 7 : 6;
  q->g.nof_re = 12 * nof_prb;
  q->g.nsym   = 2 * nsym_slot;
  q->g.sf_len = 15 * N;
  q->g.dc     = 1; // srslte_dft_plan_set_dc(true), ofdm.c:111
  q->g.norm   = 1.0f;
  q->g.cp_max = 0;
  q->g.sym0   = 0;
  q->mbsfn    = false;
  int pos = 0;
  for (int s = 0; s < q->g.nsym; s++) {
    const int cp   = cp_is_norm ? lte_cp_len_norm(s % nsym_slot, N) : lte_cp_len_ext(N);
    q->g.sym_off[s] = pos;
    q->g.cp_len[s]  = cp;
    q->g.cp_max     = cp > q->g.cp_max ? cp : q->g.cp_max;
    pos += cp + N;
  }
  q->d_shift      = nullptr;
  q->is_rx        = is_rx != 0;
  q->normalize    = false;
  q->freq_shift_f = 0.f;
  q->nof_prb      = nof_prb;
  return q;
}

extern "C" int srslte_hip_ofdm_set_normalize(srslte_hip_ofdm_t* q, int enable)
{
  if (!q) return SRSLTE_ERROR_INVALID_INPUTS;
  q->normalize = enable != 0;
  q->g.norm    = enable ? 1.0f / sqrtf((float)q->g.f.N) : 1.0f;
  q->gm.norm   = q->g.norm;
  return SRSLTE_SUCCESS;
}

extern "C" int srslte_hip_ofdm_set_mbsfn(srslte_hip_ofdm_t* q, int enable, int non_mbsfn_region)
{ // ofdm.c:424-437 (rx) and :558-574 (tx): slot 0 = `region` normal-CP symbols, a guard, then extended-CP symbols;
  // slot 1 = plain extended-CP slot (ofdm.c:453-466,:580-594). Only meaningful on an extended-CP object (6 symbols/slot).
  if (!q) return SRSLTE_ERROR_INVALID_INPUTS;
  if (!enable) {
    q->mbsfn = false;
    return SRSLTE_SUCCESS;
  }
  if (q->g.nsym != 12 || non_mbsfn_region < 1 || non_mbsfn_region > 2) {
    hip_log("[srslte_hip] MBSFN layout needs an extended-CP object and non_mbsfn_region 1 or 2 (got %d symbols, region %d)\n", q->g.nsym,
            non_mbsfn_region);
    return SRSLTE_ERROR;
  }
  if (q->d_shift) {
    hip_log("[srslte_hip] MBSFN layout with a frequency shift is not supported\n");
    return SRSLTE_ERROR;
  }
  const int N = q->g.f.N, ext = lte_cp_len_ext(N);
  q->gm       = q->g;
  int guard = non_mbsfn_region == 1 ? ext - lte_cp_len_norm(0, N) : 2 * ext - lte_cp_len_norm(0, N) - lte_cp_len_norm(1, N); // phy_common.h:147
  int pos   = 0;
  for (int i = 0; i < 6; i++) {
    if (i == non_mbsfn_region) pos += guard;
    const int cp     = i >= non_mbsfn_region ? ext : lte_cp_len_norm(i, N);
    q->gm.sym_off[i] = pos;
    q->gm.cp_len[i]  = cp;
    pos += cp + N;
  }
  q->mbsfn = true;
  return SRSLTE_SUCCESS;
}

extern "C" int srslte_hip_ofdm_set_freq_shift(srslte_hip_ofdm_t* q, float freq_shift)
{ // ofdm.c:360-378: builds the shift table and disables DC handling
  if (!q) return SRSLTE_ERROR_INVALID_INPUTS;
  if (q->mbsfn) {
    hip_log("[srslte_hip] MBSFN layout with a frequency shift is not supported\n");
    return SRSLTE_ERROR;
  }
  const int         N = q->g.f.N, len = q->g.cp_max + N;
  std::vector<cf32> tab(len);
  for (int i = 0; i < len; i++) {
    const double ph = 2.0 * M_PI * ((float)(i - q->g.cp_max)) * freq_shift / N;
    tab[i]          = make_float2((float)cos(ph), (float)sin(ph));
  }
  if (!q->d_shift) HIP_TRY(hipMalloc((void**)&q->d_shift, sizeof(cf32) * len));
  HIP_TRY(hipMemcpy(q->d_shift, tab.data(), sizeof(cf32) * len, hipMemcpyHostToDevice));
  q->g.dc         = 0;
  q->freq_shift_f = freq_shift;
  return SRSLTE_SUCCESS;
}

extern "C" void srslte_hip_ofdm_destroy(srslte_hip_ofdm_t* q)
{
  if (!q) return;
  if (q->d_shift) (void)hipFree(q->d_shift);
  delete q;
}

static int ofdm_launch(srslte_hip_ofdm_t* q, const void* d_in, void* d_out, int nof_sf, int sym0, int nsym, bool mbsfn_layout, void* stream)
{
  if (nof_sf == 0 || nsym == 0) return SRSLTE_SUCCESS;
  OfdmGeom g = mbsfn_layout ? q->gm : q->g;
  g.sym0     = sym0;
  dim3 grid(nsym, nof_sf);
  if (q->is_rx) {
    FFT_DISPATCH(ofdm_rx_kernel, g.f.N, grid, dim3(fft_threads(g.f.N)), fft_lds_bytes(g.f), (hipStream_t)stream, (const cf32*)d_in,
                 (cf32*)d_out, g, q->d_tw, (const cf32*)q->d_shift);
  } else {
    FFT_DISPATCH(ofdm_tx_kernel, g.f.N, grid, dim3(fft_threads(g.f.N)), fft_lds_bytes(g.f), (hipStream_t)stream, (const cf32*)d_in,
                 (cf32*)d_out, g, q->d_tw, (const cf32*)q->d_shift);
  }
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}

extern "C" int srslte_hip_ofdm_rx_sf_batch(srslte_hip_ofdm_t* q, const void* d_in_time, void* d_out_grid, int nof_sf, void* stream)
{
  if (!q || !d_in_time || !d_out_grid || nof_sf < 0 || !q->is_rx) return SRSLTE_ERROR_INVALID_INPUTS;
  return ofdm_launch(q, d_in_time, d_out_grid, nof_sf, 0, q->g.nsym, q->mbsfn, stream);
}

extern "C" int srslte_hip_ofdm_tx_sf_batch(srslte_hip_ofdm_t* q, const void* d_in_grid, void* d_out_time, int nof_sf, void* stream)
{
  if (!q || !d_in_grid || !d_out_time || nof_sf < 0 || q->is_rx) return SRSLTE_ERROR_INVALID_INPUTS;
  return ofdm_launch(q, d_in_grid, d_out_time, nof_sf, 0, q->g.nsym, q->mbsfn, stream);
}

extern "C" int srslte_hip_ofdm_slot_batch(srslte_hip_ofdm_t* q, const void* d_in, void* d_out, int nof_sf, int slot_in_sf, int mbsfn_layout,
                                          void* stream)
{ // one slot of every subframe; pointers are subframe bases with the subframe strides of the _sf_batch calls
  if (!q || !d_in || !d_out || nof_sf < 0 || slot_in_sf < 0 || slot_in_sf > 1 || (mbsfn_layout && !q->mbsfn)) return SRSLTE_ERROR_INVALID_INPUTS;
  return ofdm_launch(q, d_in, d_out, nof_sf, slot_in_sf * q->g.nsym / 2, q->g.nsym / 2, mbsfn_layout != 0, stream);
}

extern "C" int srslte_hip_ofdm_symbol_sz(const srslte_hip_ofdm_t* q) { return q ? q->g.f.N : -1; }
extern "C" int srslte_hip_ofdm_sf_len(const srslte_hip_ofdm_t* q) { return q ? q->g.sf_len : -1; }

// ---------------------------------------------------------------- generic DFT + transform precoding
extern "C" int srslte_hip_dft_batch(const void* d_in, void* d_out, int N, int howmany, int idist, int odist, int forward, float scale,
                                    void* stream)
{
  if (!d_in || !d_out || howmany < 0 || idist < N || odist < N) return SRSLTE_ERROR_INVALID_INPUTS;
  if (howmany == 0) return SRSLTE_SUCCESS;
  FftFactors  f;
  const cf32* d_tw;
  int         r = fft_get_plan(N, &f, &d_tw);
  if (r) return r;
  if (f.nf == 0) {
    hipLaunchKernelGGL(dft_direct_kernel, dim3((N + 255) / 256, howmany), dim3(256), 0, (hipStream_t)stream, (const cf32*)d_in, (cf32*)d_out, N, idist, odist,
                       forward ? -1.0f : 1.0f, scale, d_tw);
    LAUNCH_CHECK();
    return SRSLTE_SUCCESS;
  }
  FFT_DISPATCH(dft_batch_kernel, N, dim3(howmany), dim3(fft_threads(N)), fft_lds_bytes(f), (hipStream_t)stream, (const cf32*)d_in,
               (cf32*)d_out, f, idist, odist, forward ? -1.0f : 1.0f, scale, d_tw);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}

extern "C" int srslte_hip_dft_precoding_valid_prb(uint32_t nof_prb)
{ // dft_precoding.c:88-98
  if (nof_prb == 0 || nof_prb > 110) return 0;
  uint32_t n = nof_prb;
  while (n % 2 == 0) n /= 2;
  while (n % 3 == 0) n /= 3;
  while (n % 5 == 0) n /= 5;
  return n == 1;
}

extern "C" int srslte_hip_dft_precoding_batch(const void* d_in, void* d_out, uint32_t nof_prb, uint32_t nof_symbols, int forward, void* stream)
{ // dft_precoding.c:100-113: nof_symbols DFTs of 12*nof_prb points, 1/sqrt(N)
  if (!srslte_hip_dft_precoding_valid_prb(nof_prb)) {
    hip_log("[srslte_hip] Error invalid number of PRB (%u)\n", nof_prb);
    return SRSLTE_ERROR;
  }
  const int N = 12 * (int)nof_prb;
  return srslte_hip_dft_batch(d_in, d_out, N, (int)nof_symbols, N, N, forward, 1.0f / sqrtf((float)N), stream);
}
