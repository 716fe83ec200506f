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

constexpr int FFT_THREADS = 256;

__device__ __forceinline__ cf32 cmul(cf32 a, cf32 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cf32 cadd(cf32 a, cf32 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf32 csub(cf32 a, cf32 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * (j*s)
__device__ __forceinline__ cf32 cmulj(cf32 a, float s) { return make_float2(-s * a.y, s * a.x); }

template <int R>
__device__ __forceinline__ void butterfly(cf32* v, float sgn)
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
  } else { // 5
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
  }
}

// One Stockham pass of radix R over the whole N-point sequence held in LDS: src -> dst.
// tw[k] = exp(-j 2 pi k / N); the backward transform conjugates it.
template <int R>
__device__ __forceinline__ void stockham_pass(const cf32* __restrict__ src, cf32* __restrict__ dst, int N, int Ns,
                                              const cf32* __restrict__ tw, float sgn)
{
  const int nb   = N / R;
  const int tstep = N / (Ns * R);
  for (int j = threadIdx.x; j < nb; j += blockDim.x) {
    const int k = j % Ns;
    cf32      v[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      cf32 x = src[j + r * nb];
      if (r == 0 || k == 0) {
        v[r] = x;
      } else {
        cf32 w = tw[r * k * tstep]; // < N because k < Ns, r < R
        w.y    = -sgn * w.y; // forward (sgn=-1): table as is; backward: conjugate
        v[r]   = cmul(x, w);
      }
    }
    butterfly<R>(v, sgn);
    const int j0 = (j / Ns) * Ns * R + k;
#pragma unroll
    for (int r = 0; r < R; r++) dst[j0 + r * Ns] = v[r];
  }
}

// Runs all passes; returns the LDS buffer that holds the result.
__device__ __forceinline__ cf32* fft_in_lds(cf32* a, cf32* b, const FftFactors& f, const cf32* __restrict__ tw, float sgn)
{
  int Ns = 1;
  for (int i = 0; i < f.nf; i++) {
    __syncthreads();
    switch (f.radix[i]) {
      case 4: stockham_pass<4>(a, b, f.N, Ns, tw, sgn); break;
      case 2: stockham_pass<2>(a, b, f.N, Ns, tw, sgn); break;
      case 3: stockham_pass<3>(a, b, f.N, Ns, tw, sgn); break;
      default: stockham_pass<5>(a, b, f.N, Ns, tw, sgn); break;
    }
    Ns *= f.radix[i];
    cf32* t = a;
    a       = b;
    b       = t;
  }
  __syncthreads();
  return a;
}

struct OfdmGeom {
  FftFactors f;
  int        nof_re, nsym, sf_len, dc, cp_max;
  float      norm; // 1 or 1/sqrt(N)
  int        sym_off[14]; // first sample (start of CP) of each symbol in the subframe
  int        cp_len[14];
};

// grid = (nsym, nof_sf). in: [nof_sf][sf_len] time samples; out: [nof_sf][nsym][nof_re] resource grid.
__global__ __launch_bounds__(FFT_THREADS) void ofdm_rx_kernel(const cf32* __restrict__ in, cf32* __restrict__ out, OfdmGeom g,
                                                              const cf32* __restrict__ tw, const cf32* __restrict__ shift)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  cf32*     a   = reinterpret_cast<cf32*>(lds_raw);
  cf32*     b   = a + g.f.N;
  const int s = blockIdx.x, sf = blockIdx.y, N = g.f.N;
  const cf32* src = in + (size_t)sf * g.sf_len + g.sym_off[s] + g.cp_len[s];
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    cf32 v = src[n];
    if (shift) v = cmul(v, shift[g.cp_max + n]); // ofdm.c:369-371 with t - cplen = n
    a[n] = v;
  }
  cf32* r   = fft_in_lds(a, b, g.f, tw, -1.0f);
  cf32* dst = out + ((size_t)sf * g.nsym + s) * g.nof_re;
  const int half = g.nof_re / 2;
  for (int i = threadIdx.x; i < g.nof_re; i += blockDim.x) {
    cf32 v = i < half ? r[N - half + i] : r[g.dc + i - half]; // ofdm.c:411-412
    dst[i] = make_float2(v.x * g.norm, v.y * g.norm);
  }
}

// in: [nof_sf][nsym][nof_re] grid; out: [nof_sf][sf_len] time samples with CP.
__global__ __launch_bounds__(FFT_THREADS) void ofdm_tx_kernel(const cf32* __restrict__ in, cf32* __restrict__ out, OfdmGeom g,
                                                              const cf32* __restrict__ tw, const cf32* __restrict__ shift)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  cf32*     a   = reinterpret_cast<cf32*>(lds_raw);
  cf32*     b   = a + g.f.N;
  const int s = blockIdx.x, sf = blockIdx.y, N = g.f.N, half = g.nof_re / 2;
  const cf32* src = in + ((size_t)sf * g.nsym + s) * g.nof_re;
  for (int n = threadIdx.x; n < N; n += blockDim.x) { // ofdm.c:509-515 (guards and DC stay zero)
    cf32 v = make_float2(0.f, 0.f);
    if (n >= g.dc && n < g.dc + half) {
      v = src[half + n - g.dc];
    } else if (n >= N - half) {
      v = src[n - (N - half)];
    }
    a[n] = v;
  }
  cf32* r   = fft_in_lds(a, b, g.f, tw, 1.0f);
  cf32* dst = out + (size_t)sf * g.sf_len + g.sym_off[s];
  const int cp = g.cp_len[s];
  for (int t = threadIdx.x; t < cp + N; t += blockDim.x) { // ofdm.c:519-529: body then CP = tail copy
    cf32 v = t < cp ? r[N - cp + t] : r[t - cp];
    v      = make_float2(v.x * g.norm, v.y * g.norm);
    if (shift) v = cmul(v, shift[g.cp_max + t - cp]);
    dst[t] = v;
  }
}

// Generic batched c2c transform: howmany transforms, element strides 1, distances idist/odist, output * scale.
__global__ __launch_bounds__(FFT_THREADS) void dft_batch_kernel(const cf32* __restrict__ in, cf32* __restrict__ out, FftFactors f,
                                                                int idist, int odist, float sgn, float scale,
                                                                const cf32* __restrict__ tw)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  cf32*       a   = reinterpret_cast<cf32*>(lds_raw);
  cf32*       b   = a + f.N;
  const cf32* src = in + (size_t)blockIdx.x * idist;
  for (int n = threadIdx.x; n < f.N; n += blockDim.x) a[n] = src[n];
  cf32* r   = fft_in_lds(a, b, f, tw, sgn);
  cf32* dst = out + (size_t)blockIdx.x * odist;
  for (int n = threadIdx.x; n < f.N; n += blockDim.x) dst[n] = make_float2(r[n].x * scale, r[n].y * scale);
}

// ---------------------------------------------------------------- host side: twiddle cache
struct TwEntry {
  FftFactors f;
  cf32*      d_tw;
};
std::mutex             g_tw_mutex;
std::map<long, TwEntry> g_tw_cache; // key: device*65536 + N

int factorize(int N, FftFactors* f)
{
  f->N  = N;
  f->nf = 0;
  int n = N;
  while (n % 4 == 0 && f->nf < 8) { f->radix[f->nf++] = 4; n /= 4; }
  while (n % 2 == 0 && f->nf < 8) { f->radix[f->nf++] = 2; n /= 2; }
  while (n % 3 == 0 && f->nf < 8) { f->radix[f->nf++] = 3; n /= 3; }
  while (n % 5 == 0 && f->nf < 8) { f->radix[f->nf++] = 5; n /= 5; }
  return n == 1 ? 0 : -1;
}

} // namespace

int fft_get_plan(int N, FftFactors* f, const cf32** d_tw)
{
  if (N < 2 || N > 2048) {
    fprintf(stderr, "[srslte_hip] unsupported DFT size %d (2..2048, factors 2/3/5)\n", N);
    return SRSLTE_ERROR_INVALID_INPUTS;
  }
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_tw_mutex); // planning is serialised, like dft_fftw.c:42
  auto it = g_tw_cache.find((long)dev * 65536 + N);
  if (it == g_tw_cache.end()) {
    TwEntry e;
    if (factorize(N, &e.f)) {
      fprintf(stderr, "[srslte_hip] DFT size %d has a prime factor other than 2, 3, 5\n", N);
      return SRSLTE_ERROR_INVALID_INPUTS;
    }
    std::vector<cf32> tw(N);
    for (int k = 0; k < N; k++) tw[k] = make_float2((float)cos(2.0 * M_PI * k / N), (float)-sin(2.0 * M_PI * k / N));
    HIP_TRY(hipMalloc((void**)&e.d_tw, sizeof(cf32) * N));
    HIP_TRY(hipMemcpy(e.d_tw, tw.data(), sizeof(cf32) * N, hipMemcpyHostToDevice));
    it = g_tw_cache.emplace((long)dev * 65536 + N, e).first;
  }
  *f    = it->second.f;
  *d_tw = it->second.d_tw;
  return SRSLTE_SUCCESS;
}

// ---------------------------------------------------------------- OFDM object
struct srslte_hip_ofdm {
  OfdmGeom    g;
  const cf32* d_tw;
  cf32*       d_shift; // [cp_max + N], exp(j 2 pi m f / N), m = -cp_max .. N-1; nullptr when no shift
  bool        is_rx, normalize;
  float       freq_shift_f;
  int         nof_prb;
};

extern "C" srslte_hip_ofdm_t* srslte_hip_ofdm_create(int nof_prb, int cp_is_norm, int is_rx)
{
  const int N = lte_symbol_sz(nof_prb);
  if (N < 0) {
    fprintf(stderr, "[srslte_hip] Error: Invalid nof_prb=%d\n", nof_prb);
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
  return SRSLTE_SUCCESS;
}

extern "C" int srslte_hip_ofdm_set_freq_shift(srslte_hip_ofdm_t* q, float freq_shift)
{ // ofdm.c:360-378: builds the shift table and disables DC handling
  if (!q) return SRSLTE_ERROR_INVALID_INPUTS;
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

extern "C" int srslte_hip_ofdm_rx_sf_batch(srslte_hip_ofdm_t* q, const void* d_in_time, void* d_out_grid, int nof_sf, void* stream)
{
  if (!q || !d_in_time || !d_out_grid || nof_sf < 0 || !q->is_rx) return SRSLTE_ERROR_INVALID_INPUTS;
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  dim3 grid(q->g.nsym, nof_sf);
  hipLaunchKernelGGL(ofdm_rx_kernel, grid, dim3(FFT_THREADS), 2 * sizeof(cf32) * q->g.f.N, (hipStream_t)stream,
                     (const cf32*)d_in_time, (cf32*)d_out_grid, q->g, q->d_tw, (const cf32*)q->d_shift);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}

extern "C" int srslte_hip_ofdm_tx_sf_batch(srslte_hip_ofdm_t* q, const void* d_in_grid, void* d_out_time, int nof_sf, void* stream)
{
  if (!q || !d_in_grid || !d_out_time || nof_sf < 0 || q->is_rx) return SRSLTE_ERROR_INVALID_INPUTS;
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  dim3 grid(q->g.nsym, nof_sf);
  hipLaunchKernelGGL(ofdm_tx_kernel, grid, dim3(FFT_THREADS), 2 * sizeof(cf32) * q->g.f.N, (hipStream_t)stream,
                     (const cf32*)d_in_grid, (cf32*)d_out_time, q->g, q->d_tw, (const cf32*)q->d_shift);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
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
  hipLaunchKernelGGL(dft_batch_kernel, dim3(howmany), dim3(N >= 4 * FFT_THREADS ? FFT_THREADS : (N >= 256 ? 128 : 64)),
                     2 * sizeof(cf32) * N, (hipStream_t)stream, (const cf32*)d_in, (cf32*)d_out, f, idist, odist,
                     forward ? -1.0f : 1.0f, scale, d_tw);
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
    fprintf(stderr, "[srslte_hip] Error invalid number of PRB (%u)\n", nof_prb);
    return SRSLTE_ERROR;
  }
  const int N = 12 * (int)nof_prb;
  return srslte_hip_dft_batch(d_in, d_out, N, (int)nof_symbols, N, N, forward, 1.0f / sqrtf((float)N), stream);
}
