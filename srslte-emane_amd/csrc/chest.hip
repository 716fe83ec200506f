// Downlink channel estimator for gfx950: srslte_chest_dl_estimate_cfg (chest_dl.c:884-908) for FDD normal
// subframes, one tx port / one rx antenna per launch slice, batched over subframes.
//
// One workgroup per subframe fuses what the reference does in ~30 short vector calls: pilot gather + LS
// (refsignal_dl.c:275-295, chest_dl.c:689-690), RSRP/RSSI/CFO reductions (:558-596, :710-711), noise from
// pilots (:304-379 — only the last symbol's residual survives upstream's '=' at :374, so only that one is
// computed), Gauss/triangle smoothing with optional time averaging (:513-556, chest_common.c:62-88,
// convolution.c:180-218 "extrapolates extremes" variant) and linear interpolation in frequency and time
// (:415-511, interp.c:145-168,240-267). Pilot estimates stay in LDS; HBM traffic is the 4 pilot-bearing symbols
// in and the 14-symbol estimate out (store-bound, coalesced one RE per thread).
#include "common.hpp"
#include "phy_hip_internal.hpp"
#include <math.h>
#include <vector>

namespace {

constexpr int CH_THREADS = 256;
constexpr int MAX_NREF   = 220; // 2 * 110 PRB

struct ChestParams {
  int   cell_id, nof_prb, tti0;
  int   noise_alg, filter_type, interpolate_subframe, cfo_enable;
  float coef0, coef1;
  int   symbol_sz, cp1; // for CFO
  int   nof_rx;         // receive antennas: block v handles subframe v / nof_rx, antenna v % nof_rx ([sf][rx][grid] layouts)
};
struct ChestRaw { float noise, rsrp, rssi, cfo; }; // per (subframe, antenna), combined by chest_fill_res_kernel

struct ChestResDev { // mirrors the scalar tail of srslte_chest_dl_res_t (chest_dl.h:49-67) for 1 port / 1 antenna
  float noise_estimate, noise_estimate_dbm, snr_db, rsrp, rsrp_dbm, rsrq, rsrq_db, rssi_dbm, cfo, sync_error;
};

__device__ __forceinline__ cf32 c_add(cf32 a, cf32 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf32 c_sub(cf32 a, cf32 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf32 c_scale(cf32 a, float s) { return make_float2(a.x * s, a.y * s); }
__device__ __forceinline__ cf32 c_mulconj(cf32 a, cf32 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }

__device__ __forceinline__ int crs_nsymbol(int l) { return (l & 1) ? (l / 2 + 1) * 7 - 3 : (l / 2) * 7; } // refsignal_dl.c:234-249, normal CP, port<2
__device__ __forceinline__ int crs_fidx(int cell_id, int l) { return (((l & 1) ? 3 : 0) + (cell_id % 6)) % 6; } // port 0

__device__ float block_sum(float v, float* red)
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = 0;
  for (int i = 0; i < CH_THREADS / 64; i++) r += red[i];
  return r;
}

// "same" convolution with upstream's edge extrapolation (convolution.c:180-218)
__device__ __forceinline__ cf32 conv_at(const cf32* in, const float* h, int N, int M, int i)
{
  cf32 acc = make_float2(0.f, 0.f);
  const int H = M / 2;
  for (int t = 0; t < M; t++) {
    cf32 v;
    if (i < H) { // first[i + t]
      const int f = i + t;
      v = f < H ? c_sub(c_scale(in[1], (float)(2 + H - f)), c_scale(in[0], (float)(1 + H - f))) : in[f - H];
    } else if (i < N - H) {
      v = in[i - H + t];
    } else { // last[(i - (N - H)) + t]
      const int f = i - (N - H) + t;
      v = f >= M - 1 ? c_sub(c_scale(in[N - 1], (float)(2 + f - H)), c_scale(in[N - 2], (float)(1 + f - H))) : in[N - M + f + 1];
    }
    acc = c_add(acc, c_scale(v, h[t]));
  }
  return acc;
}

// srslte_interp_linear_offset (interp.c:240-267) evaluated at output index o
__device__ __forceinline__ cf32 interp_offset_at(const cf32* in, int L, int M, int off_st, int o)
{
  if (o < off_st) {
    const int j = off_st - o - 1;
    return c_sub(in[0], c_scale(c_scale(c_sub(in[1], in[0]), (float)(j + 1)), 1.0f / M));
  }
  const int i = (o - off_st) / M, j = (o - off_st) % M;
  if (i < L - 1) return c_add(in[i], c_scale(c_scale(c_sub(in[i + 1], in[i]), 1.0f / (float)M), (float)j));
  return c_add(in[L - 1], c_scale(c_scale(c_sub(in[L - 1], in[L - 2]), (float)j), 1.0f / M));
}

__global__ __launch_bounds__(CH_THREADS) void chest_dl_kernel(const cf32* __restrict__ grid, cf32* __restrict__ ce,
                                                             ChestResDev* __restrict__ res, ChestRaw* __restrict__ raw,
                                                             const cf32* __restrict__ pilots,
                                                             ChestParams p)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int P = p.nof_prb, nre = 12 * P, nref = 2 * P, npil = 4 * nref;
  cf32*  est = reinterpret_cast<cf32*>(lds_raw); // [4][nref]
  cf32*  avg = est + npil;                       // [4][nref]
  cf32*  fr  = avg + npil;                       // [4][nre], only when interpolate_subframe
  __shared__ float red[CH_THREADS / 64];
  __shared__ float filt[64];

  const int   sf     = blockIdx.x, sf_idx = (p.tti0 + sf / p.nof_rx) % 10, tid = threadIdx.x; // sf: (subframe, antenna) index
  const cf32* g      = grid + (size_t)sf * 14 * nre;
  const cf32* known  = pilots + (size_t)sf_idx * npil;

  // ---- pilots, LS, RSRP
  float acc = 0;
  for (int i = tid; i < npil; i += CH_THREADS) {
    const int l = i / nref, k = i - l * nref;
    cf32      r = g[crs_nsymbol(l) * nre + crs_fidx(p.cell_id, l) + 6 * k];
    est[i]      = c_mulconj(r, known[i]);
    acc += r.x * r.x + r.y * r.y;
  }
  const float rsrp = block_sum(acc, red) / npil;
  // ---- RSSI
  acc = 0;
  for (int i = tid; i < 4 * nre; i += CH_THREADS) {
    const int l = i / nre;
    cf32      v = g[crs_nsymbol(l) * nre + (i - l * nre)];
    acc += v.x * v.x + v.y * v.y;
  }
  const float rssi = block_sum(acc, red) / 4.0f;

  float cfo = 0;
  if (p.cfo_enable) { // chest_dl.c:573-596
    float sr = 0, si = 0;
    for (int i = tid; i < 2 * nref; i += CH_THREADS) {
      cf32 v = c_mulconj(est[i], est[i + 2 * nref]);
      sr += v.x;
      si += v.y;
    }
    sr = block_sum(sr, red);
    si = block_sum(si, red);
    const float n = (float)p.symbol_sz, ng = (float)p.cp1;
    cfo = (float)((double)(-atan2f(si, sr) * n / (7.0f * (n + ng))) / 2 / M_PI);
  }

  // ---- noise from pilots (REFS): residual of the last pilot symbol only (chest_dl.c:352-378)
  float noise = 0;
  if (p.noise_alg == 0) {
    const int   off = crs_fidx(p.cell_id, 0) < 3 ? 0 : 1;
    const cf32 *r0 = est, *r2 = est + 2 * nref, *r3 = est + 3 * nref;
    acc = 0;
    for (int k = tid; k < nref; k += CH_THREADS) {
      cf32 t = r3[k];
#pragma unroll
      for (int side = 0; side < 2; side++) {
        // neighbour rows: previous = row 2, next = 2*row2 - row0 (chest_dl.c:343-350)
        auto nb = [&](int idx) { return side == 0 ? r2[idx] : c_sub(c_scale(r2[idx], 2.0f), r0[idx]); };
        if (off == 0) {
          t = c_add(t, nb(k));
          t = c_add(t, k < nref - 1 ? nb(k + 1) : c_sub(c_scale(nb(nref - 2), 2.0f), nb(nref - 1)));
        } else {
          t = c_add(t, nb(k));
          t = c_add(t, k >= 1 ? nb(k - 1) : c_sub(c_scale(nb(0), 2.0f), nb(1)));
        }
      }
      t = c_sub(r3[k], c_scale(t, 1.0f / 5.0f));
      acc += t.x * t.x + t.y * t.y;
    }
    noise = block_sum(acc, red) / nref / 4.0f * sqrtf(5.0f);
  }

  if (ce) {
    // ---- smoothing filter taps (chest_dl.c:626-646)
    int flen = 0;
    if (tid == 0) {
      if (p.filter_type == 0) {
        const int   order = p.coef0 <= 0 ? 4 : (int)p.coef0;
        const float sd    = p.coef0 <= 0 ? noise * 200.0f : p.coef1;
        const int   len = order + 1, center = (len - 1) / 2;
        float       norm = 0;
        for (int i = 0; i < len; i++) {
          filt[i] = expf(-powf((float)(i - center), 2) / (2.0f * powf(sd, 2)));
          norm += filt[i];
        }
        for (int i = 0; i < len; i++) filt[i] *= 1.0f / norm;
      } else if (p.filter_type == 1) {
        filt[0] = p.coef0;
        filt[2] = p.coef0;
        filt[1] = 1 - 2 * p.coef0;
      }
    }
    flen = p.filter_type == 0 ? (p.coef0 <= 0 ? 5 : (int)p.coef0 + 1) : (p.filter_type == 1 ? 3 : 0);
    __syncthreads();

    const cf32* pil = est;
    if (p.filter_type != 2) { // average_pilots
      int n = nref, ns = 4;
      if (!p.interpolate_subframe) {
        const bool first_low = crs_fidx(p.cell_id, 0) < 3;
        for (int k = tid; k < nref; k += CH_THREADS) {
          cf32 a = c_add(est[k], est[2 * nref + k]), b = c_add(est[nref + k], est[3 * nref + k]);
          avg[2 * k]     = c_scale(first_low ? a : b, 2.0f / 4.0f);
          avg[2 * k + 1] = c_scale(first_low ? b : a, 2.0f / 4.0f);
        }
        __syncthreads();
        for (int k = tid; k < 2 * nref; k += CH_THREADS) est[k] = avg[k];
        n  = 2 * nref;
        ns = 1;
        __syncthreads();
      }
      for (int i = tid; i < ns * n; i += CH_THREADS) {
        const int l = i / n;
        avg[i]      = conv_at(est + l * n, filt, n, flen, i - l * n);
      }
      __syncthreads();
      pil = avg;
    }

    cf32* o = ce + (size_t)sf * 14 * nre;
    if (!p.interpolate_subframe) { // chest_dl.c:448-471
      const int off = p.cell_id % 3;
      for (int k = tid; k < nre; k += CH_THREADS) {
        const cf32 v = interp_offset_at(pil, 4 * P, 3, off, k);
#pragma unroll
        for (int l = 0; l < 14; l++) o[l * nre + k] = v;
      }
    } else { // chest_dl.c:456-495
      for (int i = tid; i < 4 * nre; i += CH_THREADS) {
        const int l = i / nre;
        fr[i]       = interp_offset_at(pil + nref * l, nref, 6, crs_fidx(p.cell_id, l), i - l * nre);
      }
      __syncthreads();
      for (int k = tid; k < nre; k += CH_THREADS) {
        const cf32 s0 = fr[k], s4 = fr[nre + k], s7 = fr[2 * nre + k], s11 = fr[3 * nre + k];
        cf32       d = c_scale(c_sub(s4, s0), 1.0f / 4), v = s0;
        o[k] = s0;
        for (int l = 1; l <= 3; l++) { v = c_add(v, d); o[l * nre + k] = v; }
        o[4 * nre + k] = s4;
        d = c_scale(c_sub(s7, s4), 1.0f / 3); v = s4;
        for (int l = 5; l <= 6; l++) { v = c_add(v, d); o[l * nre + k] = v; }
        o[7 * nre + k] = s7;
        d = c_scale(c_sub(s11, s7), 1.0f / 4); v = s7;
        for (int l = 8; l <= 10; l++) { v = c_add(v, d); o[l * nre + k] = v; }
        o[11 * nre + k] = s11;
        v = s11;
        for (int l = 12; l <= 13; l++) { v = c_add(v, d); o[l * nre + k] = v; }
      }
    }
  }

  if (tid == 0 && raw) raw[sf] = ChestRaw{noise, rsrp, rssi, cfo};
  if (tid == 0 && res && p.nof_rx == 1) { // fill_res (chest_dl.c:845-871), 1 port / 1 rx antenna
    ChestResDev r;
    r.noise_estimate     = noise;
    r.noise_estimate_dbm = (float)(10 * log10((double)noise) + 30);
    r.cfo                = cfo;
    r.rsrp               = rsrp;
    r.rsrp_dbm           = (float)(10 * log10((double)rsrp) + 30);
    r.rsrq               = P * rsrp / rssi;
    r.rsrq_db            = (float)(10 * log10((double)r.rsrq));
    r.snr_db             = (float)(10 * log10((double)(rsrp / noise)));
    r.rssi_dbm           = (float)(10 * log10((double)(4 * rssi / P / 12)) + 30);
    r.sync_error         = NAN;
    res[sf]              = r;
  }
}

// fill_res (chest_dl.c:747-871) for nof_rx > 1, one port: noise, RSSI and RSRQ averaged over the antennas; get_rsrp (:809-819)
// indexes ports with the antenna counter, so it is max(mean RSRP of port 0, 0); CFO of antenna 0
__global__ void chest_fill_res_kernel(const ChestRaw* __restrict__ raw, ChestResDev* __restrict__ res, int nof_sf, int nof_rx, int P)
{
  const int sf = blockIdx.x * blockDim.x + threadIdx.x;
  if (sf >= nof_sf) return;
  float noise = 0, rssi = 0, rsrq = 0, rsrp = 0;
  for (int a = 0; a < nof_rx; a++) {
    const ChestRaw r = raw[sf * nof_rx + a];
    noise += r.noise;
    rssi += 4 * r.rssi / P / 12;
    rsrq += P * r.rsrp / r.rssi;
    rsrp += r.rsrp;
  }
  noise /= nof_rx; rssi /= nof_rx; rsrq /= nof_rx; rsrp /= nof_rx;
  if (rsrp < 0.f) rsrp = 0.f;
  ChestResDev o;
  o.noise_estimate     = noise;
  o.noise_estimate_dbm = (float)(10 * log10((double)noise) + 30);
  o.cfo                = raw[sf * nof_rx].cfo;
  o.rsrp               = rsrp;
  o.rsrp_dbm           = (float)(10 * log10((double)rsrp) + 30);
  o.rsrq               = rsrq;
  o.rsrq_db            = (float)(10 * log10((double)rsrq));
  o.snr_db             = (float)(10 * log10((double)(rsrp / noise)));
  o.rssi_dbm           = (float)(10 * log10((double)rssi) + 30);
  o.sync_error         = NAN;
  res[sf]              = o;
}

// Gold sequence (sequence.c:48-79) and CRS values (refsignal_dl.c:66-116) — init-time host tables.
void gold(uint32_t c_init, uint32_t len, std::vector<uint8_t>& c)
{
  const uint32_t Nc = 1600;
  std::vector<uint8_t> x1(Nc + len + 31, 0), x2(Nc + len + 31, 0);
  for (int n = 0; n < 31; n++) x2[n] = (c_init >> n) & 1;
  x1[0] = 1;
  for (uint32_t n = 0; n < Nc + len; n++) {
    x1[n + 31] = (x1[n + 3] + x1[n]) & 1;
    x2[n + 31] = (x2[n + 3] + x2[n + 2] + x2[n + 1] + x2[n]) & 1;
  }
  c.resize(len);
  for (uint32_t n = 0; n < len; n++) c[n] = (x1[n + Nc] + x2[n + Nc]) & 1;
}

} // namespace

void lte_gold_sequence(uint32_t c_init, uint32_t len, std::vector<uint8_t>& c) { gold(c_init, len, c); }

struct srslte_hip_chest_dl {
  int       cell_id, nof_prb;
  cf32*     d_pilots; // [10][4][2*nof_prb], port 0
  ChestRaw* d_raw;    // per (subframe, antenna) scalars of multi-antenna calls, grown on demand
  size_t    raw_cap;
};

extern "C" srslte_hip_chest_dl_t* srslte_hip_chest_dl_create(uint32_t cell_id, uint32_t nof_prb, uint32_t nof_ports, int cp_is_norm)
{
  if (cell_id > 503 || nof_prb < 6 || nof_prb > 110 || nof_ports != 1 || !cp_is_norm) {
    fprintf(stderr, "[srslte_hip] chest_dl: unsupported cell (id=%u prb=%u ports=%u cp_norm=%d); single-port normal CP only\n", cell_id,
            nof_prb, nof_ports, cp_is_norm);
    return nullptr;
  }
  const int            nref = 2 * nof_prb, MAX_PRB = 110;
  std::vector<cf32>    pil((size_t)10 * 4 * nref);
  std::vector<uint8_t> c;
  for (uint32_t ns = 0; ns < 20; ns++) {
    for (uint32_t l = 0; l < 2; l++) {
      const uint32_t lp     = l == 0 ? 0 : 4;
      const uint32_t c_init = 1024 * (7 * (ns + 1) + lp + 1) * (2 * cell_id + 1) + 2 * cell_id + 1;
      gold(c_init, 4 * MAX_PRB, c);
      for (int i = 0; i < nref; i++) {
        const int mp = i + MAX_PRB - nof_prb;
        pil[((size_t)(ns / 2) * 4 + (ns % 2) * 2 + l) * nref + i] =
            make_float2((float)((1 - 2 * (float)c[2 * mp]) / sqrt(2.0)), (float)((1 - 2 * (float)c[2 * mp + 1]) / sqrt(2.0)));
      }
    }
  }
  auto* q     = new srslte_hip_chest_dl();
  q->cell_id  = cell_id;
  q->nof_prb  = nof_prb;
  q->d_pilots = nullptr;
  q->d_raw    = nullptr;
  q->raw_cap  = 0;
  if (hipMalloc((void**)&q->d_pilots, sizeof(cf32) * pil.size()) != hipSuccess ||
      hipMemcpy(q->d_pilots, pil.data(), sizeof(cf32) * pil.size(), hipMemcpyHostToDevice) != hipSuccess) {
    fprintf(stderr, "[srslte_hip] chest_dl: device allocation failed\n");
    delete q;
    return nullptr;
  }
  return q;
}

extern "C" void srslte_hip_chest_dl_destroy(srslte_hip_chest_dl_t* q)
{
  if (!q) return;
  if (q->d_pilots) (void)hipFree(q->d_pilots);
  if (q->d_raw) (void)hipFree(q->d_raw);
  delete q;
}

extern "C" const void* srslte_hip_chest_dl_pilots(const srslte_hip_chest_dl_t* q) { return q ? q->d_pilots : nullptr; }

// d_grid: [nof_sf][nof_rx][14][12*prb]; d_ce: same shape or NULL (measurements only); d_res: [nof_sf] srslte_hip_chest_res_t or NULL.
// Subframe b of the batch is TTI tti0 + b (sf_idx = TTI mod 10).
extern "C" int srslte_hip_chest_dl_estimate_batch_multi(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0,
                                                        const void* d_grid, void* d_ce, void* d_res, int nof_sf, int nof_rx, void* stream)
{
  if (!q || !cfg || !d_grid || nof_sf < 0 || nof_rx < 1 || nof_rx > 4) return SRSLTE_ERROR_INVALID_INPUTS;
  if (cfg->noise_alg != 0) {
    fprintf(stderr, "[srslte_hip] chest_dl: only SRSLTE_NOISE_ALG_REFS is implemented on device\n");
    return SRSLTE_ERROR;
  }
  if (cfg->filter_type == 0 && cfg->filter_coef[0] > 62) return SRSLTE_ERROR_INVALID_INPUTS;
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  ChestParams p;
  p.cell_id = q->cell_id; p.nof_prb = q->nof_prb; p.tti0 = (int)tti0;
  p.noise_alg = cfg->noise_alg; p.filter_type = cfg->filter_type; p.interpolate_subframe = cfg->interpolate_subframe ? 1 : 0;
  p.cfo_enable = cfg->cfo_estimate_enable ? 1 : 0;
  p.coef0 = cfg->filter_coef[0]; p.coef1 = cfg->filter_coef[1];
  p.symbol_sz = lte_symbol_sz(q->nof_prb);
  p.cp1 = lte_cp_len_norm(1, p.symbol_sz);
  p.nof_rx = nof_rx;
  ChestRaw* raw = nullptr;
  if (nof_rx > 1 && d_res) {
    const size_t need = (size_t)nof_sf * nof_rx;
    if (need > q->raw_cap) {
      if (q->d_raw) (void)hipFree(q->d_raw);
      q->d_raw = nullptr;
      HIP_TRY(hipMalloc((void**)&q->d_raw, sizeof(ChestRaw) * need));
      q->raw_cap = need;
    }
    raw = q->d_raw;
  }
  const int nref = 2 * q->nof_prb, nre = 12 * q->nof_prb;
  size_t lds = sizeof(cf32) * (8 * nref + (cfg->interpolate_subframe ? 4 * nre : 0));
  hipLaunchKernelGGL(chest_dl_kernel, dim3(nof_sf * nof_rx), dim3(CH_THREADS), lds, (hipStream_t)stream, (const cf32*)d_grid, (cf32*)d_ce,
                     (ChestResDev*)d_res, raw, (const cf32*)q->d_pilots, p);
  LAUNCH_CHECK();
  if (raw) {
    hipLaunchKernelGGL(chest_fill_res_kernel, dim3((nof_sf + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const ChestRaw*)raw,
                       (ChestResDev*)d_res, nof_sf, nof_rx, q->nof_prb);
    LAUNCH_CHECK();
  }
  return SRSLTE_SUCCESS;
}

// [nof_sf][nof_rx] x {noise, rsrp, rssi, cfo} of the last multi-antenna call with d_res != NULL (device memory owned by q): what
// the per-antenna fields of srslte_chest_dl_res_t are made of (chest_dl.c:860-870)
extern "C" const float* srslte_hip_chest_dl_last_raw(const srslte_hip_chest_dl_t* q) { return q ? (const float*)q->d_raw : nullptr; }

extern "C" int srslte_hip_chest_dl_estimate_batch(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0,
                                                  const void* d_grid, void* d_ce, void* d_res, int nof_sf, void* stream)
{
  return srslte_hip_chest_dl_estimate_batch_multi(q, cfg, tti0, d_grid, d_ce, d_res, nof_sf, 1, stream);
}
