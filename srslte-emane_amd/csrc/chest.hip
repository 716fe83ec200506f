// Downlink channel estimator for gfx950: srslte_chest_dl_estimate_cfg (chest_dl.c:884-908) for FDD normal
// subframes, 1-, 2- and 4-port cells x up to 4 rx antennas, one (subframe, port, antenna) per workgroup, batched over
// subframes. (The uplink estimator for the PUSCH DMRS is further down.)
//
// A workgroup fuses what the reference does in ~30 short vector calls: pilot gather + LS
// (refsignal_dl.c:275-295, chest_dl.c:689-690), RSRP/RSSI/CFO reductions (:558-596, :710-711), noise from
// pilots (:304-379 — only the last symbol's residual survives upstream's '=' at :374, so only that one is
// computed), Gauss/triangle smoothing with optional time averaging (:513-556, chest_common.c:62-88,
// convolution.c:180-218 "extrapolates extremes" variant) and linear interpolation in frequency and time
// (:415-511, interp.c:145-168,240-267), plus the optional sync-error and neighbour-cell measurements (:692-709). Pilot
// estimates stay in LDS; HBM traffic is the pilot-bearing symbols in and the 14-symbol estimate out (store-bound,
// coalesced one RE per thread). chest_fill_res_kernel combines the per-(port, antenna) scalars the way fill_res does.
#include "common.hpp"
#include "phy_hip_internal.hpp"
#include <map>
#include <math.h>
#include <vector>

namespace {

constexpr int CH_THREADS = 256;

struct ChestParams {
  int   cell_id, nof_prb, tti0;
  int   noise_alg, filter_type, interpolate_subframe, cfo_enable, sync_enable, corr_enable;
  float coef0, coef1;
  int   symbol_sz, cp1; // for CFO
  int   nof_rx;         // receive antennas, tx ports: block v = (sf * nof_ports + port) * nof_rx + antenna reads grid [sf][antenna]
  int   nof_ports;      // and writes ce [sf][port][antenna]
  int   nsl;            // symbols per slot: 7, or 6 in an extended-CP cell (grids and estimates are then [12][12 nof_prb])
  int   ce_compact;     // !interpolate_subframe only: ONE row of 12 nof_prb estimates per (subframe, port, antenna) instead of 2 nsl equal ones
  int   tdd_s6;         // TDD cell (srslte_cell_t.frame_type, srslte_tdd_config_t): -1 = FDD; else 1 if subframe 6 is a special subframe too
                        // (uplink-downlink configurations 0, 1, 2, 6; subframe 1 always is), 0 if not
  int   tdd_dw;         // DwPTS symbols of a special subframe (phy_common.c:128-135): only those carry CRS (refsignal_dl.c:162-225)
};
// pilot symbols of a port in a subframe (srslte_refsignal_cs_nof_symbols)
__device__ __forceinline__ int crs_nof_symbols(const ChestParams& p, int sf_idx, int port)
{
  const int full = port < 2 ? 4 : 2;
  if (p.tdd_s6 < 0 || !(sf_idx == 1 || (sf_idx == 6 && p.tdd_s6))) return full;
  const int t3 = p.nsl == 7 ? 12 : 10, t2 = p.nsl == 7 ? 9 : 8, t1 = p.nsl == 7 ? 5 : 4;
  if (p.tdd_dw >= t3) return full;
  if (p.tdd_dw >= t2) return port < 2 ? 3 : 2;
  if (p.tdd_dw >= t1) return port < 2 ? 2 : 1;
  return 1;
}
struct ChestRaw { float noise, rsrp, rssi, cfo, sync, corr; }; // per (subframe, port, antenna), combined by chest_fill_res_kernel

struct ChestResDev { // mirrors the scalar tail of srslte_chest_dl_res_t (chest_dl.h:49-67) for 1 port / 1 antenna
  float noise_estimate, noise_estimate_dbm, snr_db, rsrp, rsrp_dbm, rsrq, rsrq_db, rssi_dbm, cfo, sync_error;
};

__device__ __forceinline__ cf32 c_add(cf32 a, cf32 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf32 c_sub(cf32 a, cf32 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf32 c_scale(cf32 a, float s) { return make_float2(a.x * s, a.y * s); }
__device__ __forceinline__ cf32 c_mulconj(cf32 a, cf32 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }

// refsignal_dl.c:234-249, normal CP: ports 0/1 use symbols 0, 4, 7, 11, ports 2/3 symbols 1 and 8
__device__ __forceinline__ int crs_nsymbol(int l, int port, int nsl) { return port >= 2 ? 1 + nsl * l : ((l & 1) ? (l / 2 + 1) * nsl - 3 : (l / 2) * nsl); } // nsl: symbols per slot
// refsignal_dl.c:134-168: v = 0/3 alternating with the pilot symbol, the other way round for the odd port of each pair
__device__ __forceinline__ int crs_fidx(int cell_id, int l, int port) { return ((((l + port) & 1) ? 3 : 0) + (cell_id % 6)) % 6; }

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

// two sums with one pair of barriers (red: 2 x CH_THREADS / 64 floats)
__device__ void block_sum2(float v, float w, float* red, float* rv, float* rw)
{
  for (int o = 32; o > 0; o >>= 1) {
    v += __shfl_down(v, o, 64);
    w += __shfl_down(w, o, 64);
  }
  const int wv = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[wv]                   = v;
    red[CH_THREADS / 64 + wv] = w;
  }
  __syncthreads();
  float r = 0, q = 0;
  for (int i = 0; i < CH_THREADS / 64; i++) {
    r += red[i];
    q += red[CH_THREADS / 64 + i];
  }
  *rv = r;
  *rw = q;
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
                                                             const cf32* __restrict__ pilots, const cf32* __restrict__ pss,
                                                             const float* __restrict__ noise_state, ChestParams p)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int P = p.nof_prb, nre = 12 * P, nref = 2 * P;
  cf32*  est = reinterpret_cast<cf32*>(lds_raw); // [4][nref]
  cf32*  avg = est + 4 * nref;                   // [4][nref]
  cf32*  fr  = avg + 4 * nref;                   // [4][nre], only when interpolate_subframe
  __shared__ float red[2 * CH_THREADS / 64];
  __shared__ float filt[64];

  const int   sf = blockIdx.x, tid = threadIdx.x; // sf: (subframe, port, antenna) index
  const int   ant = sf % p.nof_rx, port = (sf / p.nof_rx) % p.nof_ports, sfn = sf / (p.nof_rx * p.nof_ports), sf_idx = (p.tti0 + sfn) % 10;
  const cf32* g      = grid + ((size_t)sfn * p.nof_rx + ant) * 2 * p.nsl * nre;
  const int   nsym = crs_nof_symbols(p, sf_idx, port), npil = nsym * nref; // 4 (ports 2/3: 2), fewer in a TDD special subframe
  // ports 0 and 1 share their values (refsignal_dl.c pilots[port / 2]), [10][4][nref]; ports 2 and 3 theirs, [10][2][nref] behind
  const cf32* known = port < 2 ? pilots + (size_t)sf_idx * 4 * nref : pilots + (size_t)10 * 4 * nref + (size_t)sf_idx * 2 * nref;

  // ---- smoothing filter taps (chest_dl.c:626-646): every tap by its own lane, then the normalisation in upstream's order (the taps' sum,
  //      i ascending). With a given order and width (or the triangular filter) they do not depend on anything measured here: made first,
  //      under the latency of the loads below; the barriers of the reductions order them before their use
  auto make_taps = [&](float noise_) {
    if (p.filter_type == 0) {
      const int   order = p.coef0 <= 0 ? 4 : (int)p.coef0;
      const float sd    = p.coef0 <= 0 ? noise_ * 200.0f : p.coef1;
      const int   len = order + 1, center = (len - 1) / 2;
      if (tid < 64) {
        const float raw = tid < len ? expf(-powf((float)(tid - center), 2) / (2.0f * powf(sd, 2))) : 0.f;
        float       norm = 0;
        for (int i = 0; i < len; i++) norm += __shfl(raw, i, 64);
        if (tid < len) filt[tid] = raw * (1.0f / norm);
      }
    } else if (p.filter_type == 1 && tid == 0) {
      filt[0] = p.coef0;
      filt[2] = p.coef0;
      filt[1] = 1 - 2 * p.coef0;
    }
  };
  const bool taps_early = ce && (p.filter_type == 1 || (p.filter_type == 0 && p.coef0 > 0));
  if (taps_early) make_taps(0.f);

  // ---- pilots, LS, RSRP and RSSI in ONE coalesced pass over the pilot-bearing symbols (the pilots sit on every sixth sub-carrier of exactly
  //      the symbols the RSSI is taken over: a strided gather of its own was a second trip to memory behind a barrier)
  float acc = 0, acc2 = 0;
  {
    int l = 0, k = tid;
    while (k >= nre) { k -= nre; l++; }
    constexpr int U = 5; // loads in flight per thread
    for (int base = 0; base < nsym * nre; base += U * CH_THREADS) {
      cf32 v[U];
      int  ll[U], kk[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        ll[u] = l; kk[u] = k;
        const bool in = l < nsym;
        v[u] = in ? g[crs_nsymbol(l, port, p.nsl) * nre + k] : make_float2(0.f, 0.f);
        k += CH_THREADS;
        while (k >= nre) { k -= nre; l++; }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (ll[u] >= nsym) continue;
        const float pw = v[u].x * v[u].x + v[u].y * v[u].y;
        acc2 += pw;
        const int d = kk[u] - crs_fidx(p.cell_id, ll[u], port);
        if (d >= 0 && d % 6 == 0) {
          const int i = ll[u] * nref + d / 6;
          est[i]      = c_mulconj(v[u], known[i]);
          acc += pw;
        }
      }
    }
  }
  float rsrp, rssi;
  block_sum2(acc, acc2, red, &rsrp, &rssi);
  rsrp /= npil;
  rssi /= (float)nsym;

  // ---- synchronisation error (chest_dl.c:692-703; srslte_vec_estimate_frequency, vector_simd.c:1606-1656, with exact divisions)
  float sync = NAN;
  if (p.sync_enable) {
    float sum = 0.f;
    for (int l = 0; l < nsym; l++) {
      const cf32* x = est + l * nref;
      float       ss = 0.f;
      for (int i = 1 + tid; i < nref; i += CH_THREADS) {
        const cf32 a = x[i], b = x[i - 1];
        ss += (a.x * b.y - b.x * a.y) / sqrtf((a.x * a.x + a.y * a.y) * (b.x * b.x + b.y * b.y));
      }
      ss = block_sum(ss, red);
      sum += asinf(ss / (float)(nref - 1)) / (2.0f * (float)M_PI) * ((float)p.symbol_sz / 6.0f);
    }
    sync = sum / (float)nsym;
  }
  // ---- power of the coherent pilot mean, for neighbour-cell RSRP (chest_dl.c:706-709)
  float corr = 0.f;
  if (p.corr_enable) {
    float sr = 0.f, si = 0.f;
    for (int i = tid; i < npil; i += CH_THREADS) {
      sr += est[i].x;
      si += est[i].y;
    }
    sr = block_sum(sr, red) / npil;
    si = block_sum(si, red) / npil;
    const double energy = sqrt((double)sr * sr + (double)si * si);
    corr                = (float)(energy * energy);
  }
  float cfo = 0;
  if (p.cfo_enable) { // chest_dl.c:573-596
    float sr = 0, si = 0;
    for (int i = tid; i < 2 * nref; i += CH_THREADS) {
      cf32 second; // chest_estimate_cfo pairs the two halves of q->pilot_estimates whatever the port (chest_dl.c:582-590): for ports 2/3
      if (port < 2) { // the second half is what port 1 of the same antenna left there: its LS estimates of symbols 7 and 11
        second = est[i + 2 * nref];
      } else {
        const int l = 2 + i / nref, k = i % nref;
        second = c_mulconj(g[crs_nsymbol(l, 1, p.nsl) * nre + crs_fidx(p.cell_id, l, 1) + 6 * k], pilots[(size_t)sf_idx * 4 * nref + l * nref + k]);
      }
      cf32 v = c_mulconj(est[i], second);
      sr += v.x;
      si += v.y;
    }
    sr = block_sum(sr, red);
    si = block_sum(si, red);
    const float n = (float)p.symbol_sz, ng = (float)p.cp1;
    cfo = (float)((double)(-atan2f(si, sr) * n / ((float)p.nsl * (n + ng))) / 2 / M_PI);
  }

  // ---- noise from pilots (REFS): residual of the last pilot symbol only (chest_dl.c:352-378); PSS / EMPTY: the estimator's kept
  // estimate [port][antenna], renewed below in subframes 0 and 5 once ce is there (:657-672)
  float noise = p.noise_alg == 0 ? 0.f : noise_state[sf % (p.nof_rx * p.nof_ports)];
  if (p.noise_alg == 0 && nsym == 1) { // "Special case for 1 symbol" (chest_dl.c:322-331): against the mean of the pilot and its two neighbours
    acc = 0;
    for (int k = tid; k + 2 < nref; k += CH_THREADS) {
      const cf32 t = c_sub(est[k + 1], c_scale(c_add(c_add(est[k + 1], est[k]), est[k + 2]), 1.0f / 3.0f));
      acc += t.x * t.x + t.y * t.y;
    }
    noise = block_sum(acc, red) / (float)(nref - 2);
  } else if (p.noise_alg == 0) {
    const int   off = ((crs_fidx(p.cell_id, 0, port) < 3) != ((nsym & 1) != 0)) ? 0 : 1; // the LAST row's offset (:353): ((fidx < 3) ^ (i & 1)) with i = nsym
    // the last pilot row, its predecessor, and the row before that (4 symbols: rows 3, 2, 0; 2 symbols: rows 1, 0, -)
    const cf32 *r0 = est, *r2 = est + (nsym - 2) * nref, *r3 = est + (nsym - 1) * nref;
    acc = 0;
    for (int k = tid; k < nref; k += CH_THREADS) {
      cf32 t = r3[k];
#pragma unroll
      for (int side = 0; side < 2; side++) {
        // neighbour rows: previous = the row before, next = its linear extrapolation 2*row2 - row0 (chest_dl.c:343-350) or, with only two
        // pilot symbols, a copy of it (:346-348)
        auto nb = [&](int idx) { return (side == 0 || nsym < 4) ? r2[idx] : c_sub(c_scale(r2[idx], 2.0f), r0[idx]); };
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
    noise = block_sum(acc, red) / nref / (float)nsym * sqrtf(5.0f);
  }

  // ---- fill_res (chest_dl.c:845-871), 1 port / 1 rx antenna, as soon as its inputs are there (res is only given with the REFS noise
  //      algorithm, whose estimate is final here): the five double-precision logarithms by five lanes of the LAST wavefront, side by side,
  //      while the first one goes on to the filter taps - one lane doing them at the kernel's end was 2 us of its 17
  if (res && p.nof_rx * p.nof_ports == 1 && tid >= CH_THREADS - 64) {
    const int   j    = tid - (CH_THREADS - 64);
    const float rsrq = P * rsrp / rssi;
    const float arg  = j == 0 ? noise : (j == 1 ? rsrp : (j == 2 ? rsrq : (j == 3 ? rsrp / noise : 4 * rssi / P / 12)));
    const float lg   = (j == 0 || j == 1 || j == 4) ? (float)(10 * log10((double)arg) + 30) : (float)(10 * log10((double)arg));
    ChestResDev r;
    r.noise_estimate     = noise;
    r.noise_estimate_dbm = __shfl(lg, 0, 64);
    r.cfo                = cfo;
    r.rsrp               = rsrp;
    r.rsrp_dbm           = __shfl(lg, 1, 64);
    r.rsrq               = rsrq;
    r.rsrq_db            = __shfl(lg, 2, 64);
    r.snr_db             = __shfl(lg, 3, 64);
    r.rssi_dbm           = __shfl(lg, 4, 64);
    r.sync_error         = sync;
    if (j == 0) res[sf] = r;
  }

  if (ce) {
    // ---- smoothing filter taps (chest_dl.c:626-646)
    // smoothing filter taps: made at the top of the kernel unless they depend on the noise estimate (automatic Gauss filter)
    const int flen = p.filter_type == 0 ? (p.coef0 <= 0 ? 5 : (int)p.coef0 + 1) : (p.filter_type == 1 ? 3 : 0);
    if (!taps_early) {
      make_taps(noise);
      __syncthreads();
    }

    const cf32* pil = est;
    if (p.filter_type != 2) { // average_pilots
      int n = nref, ns = 4;
      if (!p.interpolate_subframe && nsym > 1) { // with three rows only the first two are summed, yet scaled by 2 / 3 (chest_dl.c:527-545)
        const bool first_low = crs_fidx(p.cell_id, 0, port) < 3;
        for (int k = tid; k < nref; k += CH_THREADS) {
          cf32 a = est[k], b = est[nref + k];
          if (nsym == 4) {
            a = c_add(a, est[2 * nref + k]);
            b = c_add(b, est[3 * nref + k]);
          }
          avg[2 * k]     = c_scale(first_low ? a : b, 2.0f / (float)nsym);
          avg[2 * k + 1] = c_scale(first_low ? b : a, 2.0f / (float)nsym);
        }
        __syncthreads();
        // the one time-averaged row (2 nref values) sits in the lower half of avg, its smoothed version goes to the upper half
        for (int i = tid; i < 2 * nref; i += CH_THREADS) avg[2 * nref + i] = conv_at(avg, filt, 2 * nref, flen, i);
        __syncthreads();
        pil = avg + 2 * nref;
      } else {
        for (int i = tid; i < ns * n; i += CH_THREADS) {
          const int l = i / n;
          avg[i]      = conv_at(est + l * n, filt, n, flen, i - l * n);
        }
        __syncthreads();
        pil = avg;
      }
    }

    const int rows = p.ce_compact ? 1 : 2 * p.nsl;
    cf32*     o    = ce + (size_t)sf * rows * nre;
    if (port >= 2 && (p.interpolate_subframe || nsym == 1)) { // (nsym == 1: a special subframe's single row goes to symbol 1, then symbol 0 is copied over it)
      // ports 2/3 have two pilot symbols: upstream takes the copy branch (nsymbols < 3, chest_dl.c:467-471) and replicates symbol 0 of ce -
      // which this call does not write for them - over the subframe. ce is in / out here exactly as there.
      for (int k = tid; k < nre; k += CH_THREADS) {
        const cf32 v = o[k];
        for (int l = 1; l < 2 * p.nsl; l++) o[l * nre + k] = v;
      }
    } else if (nsym < 3 && (p.interpolate_subframe || nsym == 1)) {
      // a special subframe's one pilot row, or its two with interpolate_subframe (chest_dl.c:433,:456-471): interpolated in frequency, then
      // symbol 0 is what every symbol of the subframe gets
      for (int k = tid; k < nre; k += CH_THREADS) {
        const cf32 v = interp_offset_at(pil, nref, 6, crs_fidx(p.cell_id, 0, port), k);
#pragma unroll
        for (int l = 0; l < 14; l++) {
          if (l < rows) o[l * nre + k] = v;
        }
      }
    } else if (!p.interpolate_subframe) { // chest_dl.c:448-471
      const int off = p.cell_id % 3;
      for (int k = tid; k < nre; k += CH_THREADS) {
        const cf32 v = interp_offset_at(pil, 4 * P, 3, off, k);
#pragma unroll
        for (int l = 0; l < 14; l++) {
          if (l < rows) o[l * nre + k] = v;
        }
      }
    } else { // chest_dl.c:456-495
      for (int i = tid; i < nsym * nre; i += CH_THREADS) {
        const int l = i / nre;
        fr[i]       = interp_offset_at(pil + nref * l, nref, 6, crs_fidx(p.cell_id, l, port), i - l * nre);
      }
      __syncthreads();
      for (int k = tid; k < nre; k += CH_THREADS) {
        const cf32 s0 = fr[k], s4 = fr[nre + k], s7 = fr[2 * nre + k], s11 = fr[3 * nre + k];
        if (nsym == 3) { // special subframe with three pilot symbols (normal CP; :481-488): symbols 8 .. 13 continue the 4 -> 7 slope
          cf32 d = c_scale(c_sub(s4, s0), 1.0f / 4), v = s0;
          o[k] = s0;
          for (int l = 1; l <= 3; l++) { v = c_add(v, d); o[l * nre + k] = v; }
          o[4 * nre + k] = s4;
          d = c_scale(c_sub(s7, s4), 1.0f / 3); v = s4;
          for (int l = 5; l <= 6; l++) { v = c_add(v, d); o[l * nre + k] = v; }
          o[7 * nre + k] = s7;
          v = s7;
          for (int l = 8; l <= 13; l++) { v = c_add(v, d); o[l * nre + k] = v; }
          continue;
        }
        if (p.nsl == 6) { // extended CP: pilot symbols 0, 3, 6, 9 (chest_dl.c:497-502); the last step extrapolates with the 6-9 slope
          const cf32 pil4[4] = {s0, s4, s7, s11};
          cf32       dd = make_float2(0.f, 0.f);
          for (int seg = 0; seg < 3; seg++) {
            dd     = c_scale(c_sub(pil4[seg + 1], pil4[seg]), 1.0f / 3);
            cf32 w = pil4[seg];
            o[(3 * seg) * nre + k] = w;
            for (int l = 1; l <= 2; l++) { w = c_add(w, dd); o[(3 * seg + l) * nre + k] = w; }
          }
          cf32 w = s11;
          o[9 * nre + k] = w;
          for (int l = 10; l <= 11; l++) { w = c_add(w, dd); o[l * nre + k] = w; }
          continue;
        }
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
    if (p.noise_alg != 0 && (sf_idx == 0 || sf_idx == 5)) {
      const int k_pss = (p.nsl - 1) * nre + nre / 2 - 31, k_sss = (p.nsl - 2) * nre + nre / 2 - 31;
      const int h_pss = p.ce_compact ? nre / 2 - 31 : k_pss; // the estimate of the PSS symbol: every row holds the same values
      __syncthreads(); // the estimates of symbol 6 written above, read back by other lanes
      acc = 0;
      if (p.noise_alg == 1) { // estimate_noise_pss (chest_dl.c:381-398)
        if (tid < 62) {
          const cf32 h = o[h_pss + tid], x = pss[tid], y = g[k_pss + tid];
          const cf32 d = make_float2(h.x * x.x - h.y * x.y - y.x, h.x * x.y + h.y * x.x - y.y);
          acc          = d.x * d.x + d.y * d.y;
        }
        noise = (float)((double)((float)p.nof_ports * (block_sum(acc, red) / 62)) / sqrt(2.0));
      } else { // estimate_noise_empty_sc (:401-411): 5 empty carriers either side of the SSS and the PSS
        if (tid < 20) {
          const int  r = tid / 5, base = (r < 2 ? k_sss : k_pss) + ((r & 1) ? 62 : -5);
          const cf32 y = g[base + tid % 5];
          acc          = y.x * y.x + y.y * y.y;
        }
        noise = block_sum(acc, red) / 5;
      }
    }
  }

  if (tid == 0 && raw) raw[sf] = ChestRaw{noise, rsrp, rssi, cfo, sync, corr};
}

// MBSFN subframes (chest_dl.c:718-745 with the MBSFN branches of :304-556): one workgroup per (subframe, port, antenna). The 12-symbol
// subframe carries the port's CRS in symbol 0 and the MBSFN reference signal on every second sub-carrier of symbols 2, 6, 10 (offsets
// 0, 1, 0). LS estimates est = [2P CRS | 3 x 6P MBSFN]; the CRS row is used as it is, the MBSFN rows are smoothed; frequency interpolation
// (step 6 / step 2) is evaluated per sub-carrier straight into the time interpolation 0-2, 2-6, 6-10 and the extrapolation to 11.
// noise: estimate_noise_pilots reads the 20P estimates as 3 rows of 20P/3 (CRS included) and keeps the last row's residual (:310-378).
__global__ __launch_bounds__(CH_THREADS) void chest_dl_mbsfn_kernel(const cf32* __restrict__ grid, cf32* __restrict__ ce, float* __restrict__ noise_out,
                                                                   const cf32* __restrict__ pilots, const cf32* __restrict__ mbsfn_pilots,
                                                                   ChestParams p)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int P = p.nof_prb, nre = 12 * P, ncrs = 2 * P, nmb = 6 * P, npil = 20 * P;
  cf32*  est = reinterpret_cast<cf32*>(lds_raw); // [20 P]
  cf32*  avg = est + npil;                       // [20 P]
  __shared__ float red[CH_THREADS / 64];
  __shared__ float filt[64];
  const int   sf = blockIdx.x, tid = threadIdx.x;
  const int   ant = sf % p.nof_rx, port = (sf / p.nof_rx) % p.nof_ports, sfn = sf / (p.nof_rx * p.nof_ports), sf_idx = (p.tti0 + sfn) % 10;
  const cf32* g     = grid + ((size_t)sfn * p.nof_rx + ant) * 2 * p.nsl * nre;
  const cf32* known = pilots + (size_t)sf_idx * 4 * ncrs;      // first CRS symbol of ports 0/1
  const cf32* mb    = mbsfn_pilots + (size_t)sf_idx * 3 * nmb; // [3][6 P]
  const int   fidx  = crs_fidx(p.cell_id, 0, port);

  for (int i = tid; i < npil; i += CH_THREADS) { // srslte_refsignal_mbsfn_get_sf (refsignal_dl.c:455-487) + LS (chest_dl.c:734-741)
    if (i < ncrs) {
      est[i] = c_mulconj(g[fidx + 6 * i], known[i]);
    } else {
      const int l = (i - ncrs) / nmb, k = (i - ncrs) - l * nmb;
      est[i]      = c_mulconj(g[(2 + 4 * l) * nre + (l == 1 ? 1 : 0) + 2 * k], mb[l * nmb + k]);
    }
  }
  __syncthreads();

  float noise = 0;
  if (p.noise_alg == 0) { // last of the 3 rows: odd row index and fidx(1) = 1 < 3 give offset 1; both neighbour rows are the middle row
    const int   nref = npil / 3;
    const cf32 *r2 = est + nref, *r3 = est + 2 * nref;
    float       acc = 0;
    for (int k = tid; k < nref; k += CH_THREADS) {
      const cf32 side = c_add(r2[k], k >= 1 ? r2[k - 1] : c_sub(c_scale(r2[0], 2.0f), r2[1]));
      cf32       t    = c_add(c_add(r3[k], side), side);
      t               = c_sub(r3[k], c_scale(t, 1.0f / 5.0f));
      acc += t.x * t.x + t.y * t.y;
    }
    noise = block_sum(acc, red) / nref / 3.0f * sqrtf(5.0f);
    if (tid == 0 && noise_out) noise_out[sf] = noise;
  }
  if (!ce) return;

  if (tid == 0) { // chest_dl.c:626-646
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
  const int flen = p.filter_type == 0 ? (p.coef0 <= 0 ? 5 : (int)p.coef0 + 1) : (p.filter_type == 1 ? 3 : 0);
  __syncthreads();
  const cf32* pil = est;
  if (p.filter_type != 2) { // average_pilots, MBSFN: CRS row copied, MBSFN rows smoothed (chest_dl.c:546-555)
    for (int i = tid; i < npil; i += CH_THREADS) {
      if (i < ncrs) {
        avg[i] = est[i];
      } else {
        const int l = (i - ncrs) / nmb;
        avg[i]      = conv_at(est + ncrs + l * nmb, filt, nmb, flen, (i - ncrs) - l * nmb);
      }
    }
    __syncthreads();
    pil = avg;
  }
  cf32* o = ce + (size_t)sf * 2 * p.nsl * nre;
  for (int k = tid; k < nre; k += CH_THREADS) { // interpolate_pilots, MBSFN (chest_dl.c:436-447, :474-478)
    const cf32 s0 = interp_offset_at(pil, ncrs, 6, fidx, k), s2 = interp_offset_at(pil + ncrs, nmb, 2, 0, k);
    const cf32 s6 = interp_offset_at(pil + ncrs + nmb, nmb, 2, 1, k), s10 = interp_offset_at(pil + ncrs + 2 * nmb, nmb, 2, 0, k);
    o[k]           = s0;
    o[nre + k]     = c_add(s0, c_scale(c_sub(s2, s0), 1.0f / 2));
    o[2 * nre + k] = s2;
    cf32 d = c_scale(c_sub(s6, s2), 1.0f / 4), v = s2;
    for (int l = 3; l <= 5; l++) { v = c_add(v, d); o[l * nre + k] = v; }
    o[6 * nre + k] = s6;
    d = c_scale(c_sub(s10, s6), 1.0f / 4); v = s6;
    for (int l = 7; l <= 9; l++) { v = c_add(v, d); o[l * nre + k] = v; }
    o[10 * nre + k] = s10;
    o[11 * nre + k] = c_add(s10, d);
  }
}

// PSS / EMPTY noise: one thread per (port, antenna) walks the batch in subframe order; the estimate of a subframe 0 or 5 stays for the
// subframes after it, state carries it from and to the neighbouring calls on the object (q->noise_estimate of the reference)
__global__ void chest_noise_carry_kernel(ChestRaw* __restrict__ raw, float* __restrict__ state, int nof_sf, int nslice, int tti0, int have_ce)
{
  const int s = threadIdx.x;
  if (s >= nslice) return;
  float cur = state[s];
  for (int b = 0; b < nof_sf; b++) {
    const int sf_idx = (tti0 + b) % 10;
    if (have_ce && (sf_idx == 0 || sf_idx == 5)) cur = raw[(size_t)b * nslice + s].noise;
    else raw[(size_t)b * nslice + s].noise = cur;
  }
  state[s] = cur;
}

// fill_res (chest_dl.c:747-871) for more than one (antenna, port): noise averaged over ports and antennas; RSSI and RSRQ from port 0,
// averaged over the antennas; get_rsrp (:809-819) indexes ports with the ANTENNA counter: max over i < nof_rx of the antenna-mean RSRP
// of port i (0 for a port that was never estimated); q->cfo is overwritten by every estimate in turn: the last (antenna, port) survives
__global__ void chest_fill_res_kernel(const ChestRaw* __restrict__ raw, ChestResDev* __restrict__ res, int nof_sf, int nof_rx, int nof_ports, int P)
{
  const int sf = blockIdx.x * blockDim.x + threadIdx.x;
  if (sf >= nof_sf) return;
  const ChestRaw* r = raw + (size_t)sf * nof_ports * nof_rx; // [port][antenna]
  float noise = 0, rssi = 0, rsrq = 0;
  for (int a = 0; a < nof_rx; a++) {
    float n = 0;
    for (int pt = 0; pt < nof_ports; pt++) n += r[pt * nof_rx + a].noise;
    noise += n / nof_ports;
    rssi += 4 * r[a].rssi / P / 12;
    rsrq += P * r[a].rsrp / r[a].rssi;
  }
  noise /= nof_rx; rssi /= nof_rx; rsrq /= nof_rx;
  float rsrp = -1e9f;
  for (int i = 0; i < nof_rx; i++) {
    float v = 0;
    if (i < nof_ports) {
      for (int a = 0; a < nof_rx; a++) v += r[i * nof_rx + a].rsrp;
      v /= nof_rx;
    }
    rsrp = v > rsrp ? v : rsrp;
  }
  ChestResDev o;
  o.noise_estimate     = noise;
  o.noise_estimate_dbm = (float)(10 * log10((double)noise) + 30);
  o.cfo                = r[(nof_ports - 1) * nof_rx + nof_rx - 1].cfo;
  o.rsrp               = rsrp;
  o.rsrp_dbm           = (float)(10 * log10((double)rsrp) + 30);
  o.rsrq               = rsrq;
  o.rsrq_db            = (float)(10 * log10((double)rsrq));
  o.snr_db             = (float)(10 * log10((double)(rsrp / noise)));
  o.rssi_dbm           = (float)(10 * log10((double)rssi) + 30);
  o.sync_error         = r[0].sync; // "Take only the channel used for synch" (chest_dl.c:859)
  res[sf]              = o;
}

// Gold sequence (sequence.c:48-79; fec_tables.cpp) and CRS values (refsignal_dl.c:66-116) — init-time host tables.
inline void gold(uint32_t c_init, uint32_t len, std::vector<uint8_t>& c) { lte_gold_sequence(c_init, len, c); }

} // namespace


struct srslte_hip_chest_dl {
  int       cell_id, nof_prb, nof_ports, nsl; // nsl: symbols per slot (7, extended CP 6)
  cf32*     d_pilots; // [10][4][2*nof_prb] ports 0 and 1, then [10][2][2*nof_prb] ports 2 and 3 (4-port cells)
  ChestRaw* d_raw;    // per (subframe, port, antenna) scalars of multi-antenna / multi-port calls, grown on demand
  size_t    raw_cap;
  cf32*     d_mbsfn[256]; // per MBSFN area id: [10][3][6*nof_prb] (set_mbsfn_area_id), or null
  cf32*     d_pss;        // the cell's 62 PSS values (pss.c:348-376), for the PSS noise algorithm
  float*    d_noise_state; // [port][antenna] noise estimates kept between calls by the PSS / EMPTY algorithms
  int       symbol_sz;     // srslte_symbol_sz(nof_prb) as the CFO and timing estimates use it (chest_dl.c:575,:695)
  int       tdd_s6, tdd_dw; // srslte_hip_chest_dl_set_tdd: -1 = FDD (ChestParams)
};

extern "C" srslte_hip_chest_dl_t* srslte_hip_chest_dl_create(uint32_t cell_id, uint32_t nof_prb, uint32_t nof_ports, int cp_is_norm)
{
  if (cell_id > 503 || nof_prb < 6 || nof_prb > 110 || (nof_ports != 1 && nof_ports != 2 && nof_ports != 4)) {
    hip_log("[srslte_hip] chest_dl: unsupported cell (id=%u prb=%u ports=%u cp_norm=%d); 1, 2 or 4 ports\n", cell_id,
            nof_prb, nof_ports, cp_is_norm);
    return nullptr;
  }
  const int            nref = 2 * nof_prb, MAX_PRB = 110;
  std::vector<cf32>    pil((size_t)10 * 6 * nref);
  std::vector<uint8_t> c;
  for (uint32_t ns = 0; ns < 20; ns++) {
    for (uint32_t l = 0; l < 3; l++) { // l = 0, 1: symbols 0 and 4 of the slot (ports 0/1); l = 2: symbol 1 (ports 2/3)
      const uint32_t lp     = l == 0 ? 0 : (l == 1 ? (cp_is_norm ? 4 : 3) : 1);
      const uint32_t c_init = 1024 * (7 * (ns + 1) + lp + 1) * (2 * cell_id + 1) + 2 * cell_id + (cp_is_norm ? 1 : 0); // N_cp (refsignal_dl.c:92)
      gold(c_init, 4 * MAX_PRB, c);
      cf32* dst = l < 2 ? &pil[((size_t)(ns / 2) * 4 + (ns % 2) * 2 + l) * nref] : &pil[(size_t)10 * 4 * nref + ((size_t)(ns / 2) * 2 + ns % 2) * nref];
      for (int i = 0; i < nref; i++) {
        const int mp = i + MAX_PRB - nof_prb;
        dst[i]       = make_float2((float)((1 - 2 * (float)c[2 * mp]) / sqrt(2.0)), (float)((1 - 2 * (float)c[2 * mp + 1]) / sqrt(2.0)));
      }
    }
  }
  auto* q     = new srslte_hip_chest_dl();
  q->cell_id  = cell_id;
  q->nof_prb  = nof_prb;
  q->nof_ports = (int)nof_ports;
  q->nsl      = cp_is_norm ? 7 : 6;
  q->tdd_s6   = -1;
  q->tdd_dw   = 0;
  q->d_pilots = nullptr;
  q->d_raw    = nullptr;
  q->raw_cap  = 0;
  for (auto& m : q->d_mbsfn) m = nullptr;
  q->d_pss = nullptr;
  q->d_noise_state = nullptr;
  q->symbol_sz     = lte_symbol_sz((int)nof_prb);
  cf32 pss[62];
  {
    const float root_value[] = {25.0, 29.0, 34.0};
    for (int i = 0; i < 62; i++) {
      const float arg = i < 31 ? (float)-1 * M_PI * root_value[cell_id % 3] * ((float)i * ((float)i + 1.0)) / 63.0
                               : (float)-1 * M_PI * root_value[cell_id % 3] * (((float)i + 2.0) * ((float)i + 1.0)) / 63.0;
      pss[i] = make_float2(cosf(arg), sinf(arg));
    }
  }
  if (hipMalloc((void**)&q->d_pilots, sizeof(cf32) * pil.size()) != hipSuccess ||
      hipMalloc((void**)&q->d_pss, sizeof(pss)) != hipSuccess || hipMemcpy(q->d_pss, pss, sizeof(pss), hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc((void**)&q->d_noise_state, sizeof(float) * 16) != hipSuccess || hipMemset(q->d_noise_state, 0, sizeof(float) * 16) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess /* null-stream memset vs the callers' non-blocking streams */ ||
      hipMemcpy(q->d_pilots, pil.data(), sizeof(cf32) * pil.size(), hipMemcpyHostToDevice) != hipSuccess) {
    hip_log("[srslte_hip] chest_dl: device allocation failed\n");
    delete q;
    return nullptr;
  }
  return q;
}

extern "C" int srslte_hip_chest_dl_set_symbol_sz(srslte_hip_chest_dl_t* q, int symbol_sz)
{ // chest_dl.c:575,:695 read srslte_symbol_sz(cell.nof_prb) at every call: the other rate family after srslte_use_standard_symbol_size(true)
  if (!q || symbol_sz <= 12 * q->nof_prb || symbol_sz > 2048) return SRSLTE_ERROR_INVALID_INPUTS;
  q->symbol_sz = symbol_sz;
  return SRSLTE_SUCCESS;
}

extern "C" void srslte_hip_chest_dl_destroy(srslte_hip_chest_dl_t* q)
{
  if (!q) return;
  if (q->d_pilots) (void)hipFree(q->d_pilots);
  if (q->d_raw) (void)hipFree(q->d_raw);
  if (q->d_pss) (void)hipFree(q->d_pss);
  if (q->d_noise_state) (void)hipFree(q->d_noise_state);
  for (auto m : q->d_mbsfn) {
    if (m) (void)hipFree(m);
  }
  delete q;
}

extern "C" int srslte_hip_chest_dl_set_mbsfn_area_id(srslte_hip_chest_dl_t* q, uint16_t mbsfn_area_id)
{ // srslte_chest_dl_set_mbsfn_area_id (chest_dl.c:244-262) with srslte_refsignal_mbsfn_gen_seq (refsignal_dl.c:361-400)
  if (!q || mbsfn_area_id > 255) return SRSLTE_ERROR_INVALID_INPUTS;
  if (q->d_mbsfn[mbsfn_area_id]) return SRSLTE_SUCCESS;
  const int            nmb = 6 * q->nof_prb, MAX_PRB = 110;
  std::vector<cf32>    pil((size_t)10 * 3 * nmb);
  std::vector<uint8_t> c;
  for (uint32_t sf = 0; sf < 10; sf++) {
    for (uint32_t l = 0; l < 3; l++) {
      const uint32_t lp = (2 + 4 * l) % 6, slot = l ? 2 * sf + 1 : 2 * sf;
      gold(512 * (7 * (slot + 1) + lp + 1) * (2 * (uint32_t)mbsfn_area_id + 1) + mbsfn_area_id, 20 * MAX_PRB, c);
      for (int i = 0; i < nmb; i++) {
        const int mp = i + 3 * (MAX_PRB - q->nof_prb);
        pil[((size_t)sf * 3 + l) * nmb + i] =
            make_float2((float)((1 - 2 * (float)c[2 * mp]) / sqrt(2.0)), (float)((1 - 2 * (float)c[2 * mp + 1]) / sqrt(2.0)));
      }
    }
  }
  cf32* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, sizeof(cf32) * pil.size()));
  if (hipMemcpy(d, pil.data(), sizeof(cf32) * pil.size(), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d);
    return SRSLTE_ERROR;
  }
  q->d_mbsfn[mbsfn_area_id] = d;
  return SRSLTE_SUCCESS;
}

extern "C" const void* srslte_hip_chest_dl_mbsfn_pilots(const srslte_hip_chest_dl_t* q, uint16_t mbsfn_area_id)
{
  return q && mbsfn_area_id < 256 ? q->d_mbsfn[mbsfn_area_id] : nullptr;
}

// fill_res after an MBSFN estimate (chest_dl.c:845-871): only the noise figure is new - get_noise (:747-758), the mean over the antennas of the
// mean over the ports of the REFS estimates; the other fields keep the last normal subframe's values upstream, zero here (the pipeline reads the
// noise figure only). noise: [nof_sf][nof_ports][nof_rx]
__global__ void chest_mbsfn_res_kernel(const float* __restrict__ noise, ChestResDev* __restrict__ res, int nof_sf, int nof_rx, int nof_ports)
{
  const int sf = blockIdx.x * blockDim.x + threadIdx.x;
  if (sf >= nof_sf) return;
  const float* r = noise + (size_t)sf * nof_ports * nof_rx;
  float        n = 0;
  for (int a = 0; a < nof_rx; a++) {
    float m = 0;
    for (int pt = 0; pt < nof_ports; pt++) m += r[pt * nof_rx + a];
    n += m / nof_ports;
  }
  n /= nof_rx;
  ChestResDev o = {};
  o.noise_estimate     = n;
  o.noise_estimate_dbm = (float)(10 * log10((double)n) + 30);
  res[sf]              = o;
}

static int chest_dl_mbsfn_impl(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0, const void* d_grid, void* d_ce,
                               float* d_noise, int nof_sf, int nof_rx, int nsl, void* stream);

extern "C" int srslte_hip_chest_dl_estimate_mbsfn_batch(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0,
                                                        const void* d_grid, void* d_ce, float* d_noise, int nof_sf, int nof_rx, void* stream)
{
  return chest_dl_mbsfn_impl(q, cfg, tti0, d_grid, d_ce, d_noise, nof_sf, nof_rx, q ? q->nsl : 0, stream);
}

// The PMCH pipeline's call: grids and estimates of 12 symbols per subframe whatever the cell's CP (nsl = 6), and a result record per subframe
// whose noise figure is get_noise over the antennas (REFS algorithm; with PSS / EMPTY an MBSFN subframe measures nothing upstream)
int chest_dl_estimate_mbsfn_rows(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0, const void* d_grid, void* d_ce,
                                 int nof_sf, int nof_rx, int nsl, void* d_res, void* stream)
{
  if (!q || !cfg || !d_res || nof_sf < 0) return SRSLTE_ERROR_INVALID_INPUTS;
  if (cfg->noise_alg != 0) { // with PSS / EMPTY an MBSFN subframe measures nothing (they are never subframe 0 or 5): the equaliser would get a stale figure
    hip_log("[srslte_hip] chest_dl: the MBSFN pipeline needs the REFS noise estimate\n");
    return SRSLTE_ERROR_INVALID_INPUTS;
  }
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  const size_t need = ((size_t)nof_sf * nof_rx * q->nof_ports * sizeof(float) + sizeof(ChestRaw) - 1) / sizeof(ChestRaw);
  if (need > q->raw_cap) {
    if (q->d_raw) (void)hipFree(q->d_raw);
    q->d_raw = nullptr;
    q->raw_cap = 0;
    HIP_TRY(hipMalloc((void**)&q->d_raw, sizeof(ChestRaw) * need));
    q->raw_cap = need;
  }
  float* d_noise = reinterpret_cast<float*>(q->d_raw);
  if (int rc = chest_dl_mbsfn_impl(q, cfg, tti0, d_grid, d_ce, d_noise, nof_sf, nof_rx, nsl, stream)) return rc;
  hipLaunchKernelGGL(chest_mbsfn_res_kernel, dim3(ceil_div(nof_sf, 64)), dim3(64), 0, (hipStream_t)stream, (const float*)d_noise, (ChestResDev*)d_res,
                     nof_sf, nof_rx, q->nof_ports);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}

static int chest_dl_mbsfn_impl(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0, const void* d_grid, void* d_ce,
                               float* d_noise, int nof_sf, int nof_rx, int nsl, void* stream)
{
  if (!q || !cfg || !d_grid || nof_sf < 0 || nof_rx < 1 || nof_rx > 4 || cfg->mbsfn_area_id > 255) return SRSLTE_ERROR_INVALID_INPUTS;
  if (!q->d_mbsfn[cfg->mbsfn_area_id]) {
    hip_log("[srslte_hip] chest_dl: MBSFN area id=%d not initialized\n", cfg->mbsfn_area_id); // chest_dl.c:729-731
    return SRSLTE_ERROR;
  }
  if (q->nof_ports > 2 || (!cfg->interpolate_subframe && d_ce)) {
    // upstream then interpolates in time from symbols nothing wrote (chest_dl.c:430-433,:474-478; ports 2/3 leave symbol 0 unwritten)
    hip_log("[srslte_hip] chest_dl: MBSFN subframes need interpolate_subframe and a 1- or 2-port cell\n");
    return SRSLTE_ERROR;
  }
  if (cfg->filter_type == 0 && cfg->filter_coef[0] > 62) return SRSLTE_ERROR_INVALID_INPUTS;
  if (cfg->filter_type == 0 && cfg->filter_coef[0] <= 0 && cfg->noise_alg != 0 && d_ce) {
    hip_log("[srslte_hip] chest_dl: the automatic Gauss filter needs the REFS noise estimate in an MBSFN subframe\n");
    return SRSLTE_ERROR;
  }
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  ChestParams p;
  p = ChestParams{};
  p.tdd_s6 = -1;
  p.cell_id = q->cell_id; p.nof_prb = q->nof_prb; p.tti0 = (int)tti0;
  p.noise_alg = cfg->noise_alg; p.filter_type = cfg->filter_type; p.interpolate_subframe = 1;
  p.coef0 = cfg->filter_coef[0]; p.coef1 = cfg->filter_coef[1];
  p.nof_rx = nof_rx; p.nof_ports = q->nof_ports; p.nsl = nsl;
  hipLaunchKernelGGL(chest_dl_mbsfn_kernel, dim3(nof_sf * nof_rx * q->nof_ports), dim3(CH_THREADS), sizeof(cf32) * 40 * q->nof_prb,
                     (hipStream_t)stream, (const cf32*)d_grid, (cf32*)d_ce, d_noise, (const cf32*)q->d_pilots,
                     (const cf32*)q->d_mbsfn[cfg->mbsfn_area_id], p);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}

// TDD cell (srslte_cell_t.frame_type = SRSLTE_TDD with the subframes' srslte_tdd_config_t): in special subframes (1, and 6 in uplink-downlink
// configurations 0, 1, 2, 6) only the DwPTS symbols carry CRS and the estimator works on 4, 3, 2 or 1 pilot symbols (refsignal_dl.c:162-225,
// chest_dl.c:322-331,:481-488,:527-545). sf_config < 0: back to FDD. Not with cfo_estimate_enable (upstream pairs rows a shortened subframe lacks)
// nor, on an extended-CP cell, with interpolate_subframe (upstream's TODO, chest_dl.c:497): those calls are refused.
extern "C" int srslte_hip_chest_dl_set_tdd(srslte_hip_chest_dl_t* q, int sf_config, int ss_config)
{
  static const int dw[10] = {3, 9, 10, 11, 12, 3, 9, 10, 11, 6}; // phy_common.c:98-99, first column
  if (!q || sf_config > 6 || (sf_config >= 0 && (ss_config < 0 || ss_config > 9))) return SRSLTE_ERROR_INVALID_INPUTS;
  q->tdd_s6 = sf_config < 0 ? -1 : ((sf_config <= 2 || sf_config == 6) ? 1 : 0);
  q->tdd_dw = sf_config < 0 ? 0 : dw[ss_config];
  return SRSLTE_SUCCESS;
}

int chest_dl_set_noise_state(srslte_hip_chest_dl_t* q, const float* noise /* [port][antenna], 16 */)
{
  if (!q || !noise) return SRSLTE_ERROR_INVALID_INPUTS;
  return hipMemcpy(q->d_noise_state, noise, sizeof(float) * 16, hipMemcpyHostToDevice) == hipSuccess ? SRSLTE_SUCCESS : SRSLTE_ERROR;
}

extern "C" const void* srslte_hip_chest_dl_pilots(const srslte_hip_chest_dl_t* q) { return q ? q->d_pilots : nullptr; }

// d_grid: [nof_sf][nof_rx][14][12*prb]; d_ce: [nof_sf][nof_ports][nof_rx][14][12*prb] or NULL (measurements only); d_res: [nof_sf]
// srslte_hip_chest_res_t or NULL.
// Subframe b of the batch is TTI tti0 + b (sf_idx = TTI mod 10).
extern "C" int srslte_hip_chest_dl_estimate_batch_multi(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0,
                                                        const void* d_grid, void* d_ce, void* d_res, int nof_sf, int nof_rx, void* stream)
{
  return chest_dl_estimate_batch_rows(q, cfg, tti0, d_grid, d_ce, d_res, nof_sf, nof_rx, 0, stream);
}

// ce_compact: d_ce is [nof_sf][nof_ports][nof_rx][12*prb] - without interpolate_subframe the reference copies ONE row of estimates to
// every symbol of the subframe (chest_dl.c:467-471); the fused receive pipeline keeps that row only (its demapper reads it for every symbol)
int chest_dl_estimate_batch_rows(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0, const void* d_grid, void* d_ce,
                                 void* d_res, int nof_sf, int nof_rx, int ce_compact, void* stream)
{
  if (!q || !cfg || !d_grid || nof_sf < 0 || nof_rx < 1 || nof_rx > 4) return SRSLTE_ERROR_INVALID_INPUTS;
  if (ce_compact && cfg->interpolate_subframe) return SRSLTE_ERROR_INVALID_INPUTS;
  if (q->tdd_s6 >= 0 && q->tdd_dw < (q->nsl == 7 ? 12 : 10) && (cfg->cfo_estimate_enable || (q->nsl == 6 && cfg->interpolate_subframe && d_ce))) {
    hip_log("[srslte_hip] chest_dl: TDD special subframes with fewer pilot symbols: no CFO estimate, no interpolate_subframe on an extended-CP cell\n");
    return SRSLTE_ERROR;
  }
  if (cfg->noise_alg < 0 || cfg->noise_alg > 2) return SRSLTE_ERROR_INVALID_INPUTS;
  if (cfg->noise_alg != 0 && cfg->filter_type == 0 && cfg->filter_coef[0] <= 0 && nof_sf > 1 && d_ce) {
    // the automatic Gauss filter of a subframe then depends on the estimates of the subframes before it: a sequential chain
    hip_log("[srslte_hip] chest_dl: the automatic Gauss filter with the PSS / EMPTY noise algorithms needs one subframe per call\n");
    return SRSLTE_ERROR;
  }
  if (cfg->filter_type == 0 && cfg->filter_coef[0] > 62) return SRSLTE_ERROR_INVALID_INPUTS;
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  ChestParams p;
  p.cell_id = q->cell_id; p.nof_prb = q->nof_prb; p.tti0 = (int)tti0;
  p.noise_alg = cfg->noise_alg; p.filter_type = cfg->filter_type; p.interpolate_subframe = cfg->interpolate_subframe ? 1 : 0;
  p.cfo_enable = cfg->cfo_estimate_enable ? 1 : 0;
  p.sync_enable = cfg->sync_error_enable ? 1 : 0;
  p.corr_enable = cfg->rsrp_neighbour ? 1 : 0;
  p.coef0 = cfg->filter_coef[0]; p.coef1 = cfg->filter_coef[1];
  p.symbol_sz = q->symbol_sz;
  p.cp1 = lte_cp_len_norm(1, p.symbol_sz);
  p.nof_rx = nof_rx;
  p.nof_ports = q->nof_ports;
  p.nsl = q->nsl;
  p.ce_compact = ce_compact ? 1 : 0;
  p.tdd_s6 = q->tdd_s6; p.tdd_dw = q->tdd_dw;
  const int nslice = nof_rx * q->nof_ports;
  ChestRaw* raw = nullptr;
  if (d_res || cfg->noise_alg) {
    const size_t need = (size_t)nof_sf * nslice;
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
  hipLaunchKernelGGL(chest_dl_kernel, dim3(nof_sf * nslice), dim3(CH_THREADS), lds, (hipStream_t)stream, (const cf32*)d_grid, (cf32*)d_ce,
                     cfg->noise_alg ? nullptr : (ChestResDev*)d_res, raw, (const cf32*)q->d_pilots, (const cf32*)q->d_pss,
                     (const float*)q->d_noise_state, p);
  LAUNCH_CHECK();
  if (raw && cfg->noise_alg) {
    hipLaunchKernelGGL(chest_noise_carry_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, raw, q->d_noise_state, nof_sf, nslice, (int)tti0,
                       d_ce ? 1 : 0);
    LAUNCH_CHECK();
  }
  if (raw && d_res && (nslice > 1 || cfg->noise_alg)) {
    hipLaunchKernelGGL(chest_fill_res_kernel, dim3((nof_sf + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const ChestRaw*)raw,
                       (ChestResDev*)d_res, nof_sf, nof_rx, q->nof_ports, q->nof_prb);
    LAUNCH_CHECK();
  }
  return SRSLTE_SUCCESS;
}

// [nof_sf][nof_ports][nof_rx] x {noise, rsrp, rssi, cfo, sync, corr} of the last call with d_res != NULL and more than one (port,
// antenna) or cfg.rsrp_neighbour (device memory owned by q): what
// the per-antenna fields of srslte_chest_dl_res_t are made of (chest_dl.c:860-870)
extern "C" const float* srslte_hip_chest_dl_last_raw(const srslte_hip_chest_dl_t* q) { return q ? (const float*)q->d_raw : nullptr; }

extern "C" int srslte_hip_chest_dl_estimate_batch(srslte_hip_chest_dl_t* q, const srslte_hip_chest_dl_cfg_t* cfg, uint32_t tti0,
                                                  const void* d_grid, void* d_ce, void* d_res, int nof_sf, void* stream)
{
  return srslte_hip_chest_dl_estimate_batch_multi(q, cfg, tti0, d_grid, d_ce, d_res, nof_sf, 1, stream);
}

// ====================================================================================================================
// Uplink: PUSCH DMRS (refsignal_ul.c) and srslte_chest_ul_estimate_pusch (chest_ul.c) - SURVEY §8f N3
// ====================================================================================================================
namespace {

const uint32_t N_DMRS_1[8] = {0, 2, 3, 4, 6, 8, 9, 10}; // 36.211 Table 5.5.2.1.1-2 (refsignal_ul.c:42)
const uint32_t N_DMRS_2[8] = {0, 6, 3, 4, 2, 8, 10, 9}; // 36.211 Table 5.5.2.1.1-1 (refsignal_ul.c:39)

// 36.211 Tables 5.5.1.2-1 and 5.5.1.2-2: phi(n) of the QPSK base sequences for M_sc = 12 and 24 (one and two PRB), one string per
// group u; digit d stands for phi = 2 d - 3, i.e. 0 1 2 3 = -3 -1 +1 +3 (the reference keeps them as int arrays, ul_rs_tables.h)
const char* const PHI_12[30] = {
    "123033223203",
    "223331200203",
    "220001002021",
    "122221002031",
    "132121012123",
    "203112211302",
    "130003213302",
    "011120312032",
    "203211122312",
    "201331022222",
    "131220010031",
    "321133023233",
    "202202220002",
    "330302231033",
    "021013233312",
    "312011223210",
    "232123331131",
    "022303003231",
    "032202001120",
    "132321130101",
    "102222321201",
    "131200000210",
    "220000130203",
    "221010212312",
    "223233121002",
    "203323320113",
    "230030211310",
    "010103212300",
    "130313303311",
    "300110130321",
};
const char* const PHI_24[30] = {
    "132031230323032212303010",
    "030002003122232130023220",
    "313322033332131221011233",
    "102230220112323213220101",
    "111001223313121021002011",
    "022312320202211310300022",
    "221130030211212210121310",
    "033110132323221321230112",
    "023021030311112000200020",
    "220331013033312202122022",
    "120031311000100212331213",
    "233002321000330331031202",
    "233222112031220331030101",
    "311110133212333122023103",
    "003232032322331102013223",
    "112023021013232100110001",
    "103111122032332120202201",
    "231331021033312231013111",
    "222221310223020122003220",
    "233210313330212101231300",
    "103000110103230131213021",
    "002212121320121211330120",
    "010321010030301232023310",
    "111133323302313133032133",
    "213310301131311222211013",
    "212131322110220230220011",
    "012322011030320302120222",
    "103322310111320013010101",
    "101120112102202003221311",
    "221101313123213230021123",
};

uint32_t largest_prime_below(uint32_t x)
{
  for (uint32_t p = x - 1; p >= 2; p--) {
    bool prime = true;
    for (uint32_t d = 2; d * d <= p; d++) {
      if (p % d == 0) {
        prime = false;
        break;
      }
    }
    if (prime) return p;
  }
  return 0;
}

struct ChestUlGeom {
  int   cell_nre, L_prb, n_prb, n_prb1, tti0; // 12 * cell nof_prb; grant: PRB offset of slot 0 and of slot 1 (srslte_pusch_grant_t.n_prb[2])
  int   nsl;                          // symbols per slot: 7, or 6 with the extended CP (DMRS in symbol nsl - 4 of each slot, refsignal_ul.h:43)
  float w;                            // 3-tap smoothing filter {w, 1-2w, w} (chest_ul.c:101-102)
};
struct ChestUlResDev { float noise_estimate, noise_estimate_dbm, snr, snr_db, cfo; };

// One workgroup per subframe (chest_ul.c:268-327): LS estimates at the two DMRS symbols, 3-tap "same" convolution with the
// edge extrapolation of srslte_conv_same_cf (convolution.c:180-218), the result copied to the 7 (6: extended CP) symbols of its slot
// (DO_LINEAR_INTERPOLATION is not defined upstream), noise from the difference smoothed - raw, SNR from the pilot power.
// d_r: [10][2][12 * L_prb] DMRS of the grant per subframe index.
// items != null (per-PUSCH grants): workgroup i estimates items[i] - its subframe of the batch, its PRB offsets, its result row - and all
// items of one launch share L_prb and the DMRS table.
__global__ __launch_bounds__(CH_THREADS) void chest_ul_kernel(const cf32* __restrict__ grid, cf32* __restrict__ ce,
                                                             ChestUlResDev* __restrict__ res, const cf32* __restrict__ d_r, ChestUlGeom g,
                                                             const ChestUlItem* __restrict__ items)
{
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ float red[CH_THREADS / 64];
  const int   sf = items ? items[blockIdx.x].sf : (int)blockIdx.x, row = items ? items[blockIdx.x].row : sf;
  const int   n_prb0 = items ? items[blockIdx.x].n_prb : g.n_prb, n_prb1 = items ? items[blockIdx.x].n_prb1 : g.n_prb1;
  const int   sf_idx = (g.tti0 + sf) % 10, tid = threadIdx.x, nrefs = 12 * g.L_prb;
  cf32*       est = reinterpret_cast<cf32*>(lds_raw); // [2][nrefs]
  const cf32* gs  = grid + (size_t)sf * 2 * g.nsl * g.cell_nre;
  cf32*       cs  = ce ? ce + (size_t)sf * 2 * g.nsl * g.cell_nre : nullptr;
  const cf32* r   = d_r + (size_t)sf_idx * 2 * nrefs;
  float       pw  = 0.f;
  for (int i = tid; i < 2 * nrefs; i += CH_THREADS) {
    const int  s = i / nrefs, k = i - s * nrefs, L = (s + 1) * g.nsl - 4;
    const cf32 y = gs[L * g.cell_nre + (s ? n_prb1 : n_prb0) * 12 + k];
    est[i]       = c_mulconj(y, r[i]);
    pw += y.x * y.x + y.y * y.y;
  }
  __syncthreads();
  const float pilot_power = block_sum(pw, red) / (float)(2 * nrefs);
  const float f0 = g.w, f1 = 1 - 2 * g.w;
  float       npw[2] = {0.f, 0.f};
  for (int i = tid; i < 2 * nrefs; i += CH_THREADS) {
    const int   s = i / nrefs, k = i - s * nrefs;
    const cf32* e = est + s * nrefs;
    // conv_same with M = 3: out[k] = f0 * in[k-1] + f1 * in[k] + f0 * in[k+1]; at the two ends the missing neighbour is upstream's
    // "extrapolated" value 3 * in[1] - 2 * in[0] resp. 3 * in[N-1] - 2 * in[N-2] (convolution.c:180-218, reproduced as it is)
    cf32 o;
    if (nrefs < 3) {
      o = e[k];
    } else if (k == 0) {
      const cf32 first = c_sub(c_scale(e[1], 3.0f), c_scale(e[0], 2.0f));
      o = c_add(c_add(c_scale(first, f0), c_scale(e[0], f1)), c_scale(e[1], f0));
    } else if (k == nrefs - 1) {
      const cf32 last = c_sub(c_scale(e[nrefs - 1], 3.0f), c_scale(e[nrefs - 2], 2.0f));
      o = c_add(c_add(c_scale(e[nrefs - 2], f0), c_scale(e[nrefs - 1], f1)), c_scale(last, f0));
    } else {
      o = c_add(c_add(c_scale(e[k - 1], f0), c_scale(e[k], f1)), c_scale(e[k + 1], f0));
    }
    if (cs) {
      for (int l = 0; l < g.nsl; l++) cs[(s * g.nsl + l) * g.cell_nre + (s ? n_prb1 : n_prb0) * 12 + k] = o;
    }
    const cf32 d = c_sub(o, e[k]);
    npw[s] += d.x * d.x + d.y * d.y;
  }
  const float p0 = block_sum(npw[0], red) / (float)nrefs, p1 = block_sum(npw[1], red) / (float)nrefs;
  if (tid == 0 && res) {
    const float power = (p0 + p1) / 2;
    const float a     = (float)(7.419 * g.w * g.w + 0.1117 * g.w - 0.005387); // chest_ul.c:217-221
    ChestUlResDev o;
    o.noise_estimate     = (float)(power / (a * 0.8));
    o.snr                = o.noise_estimate ? pilot_power / o.noise_estimate : NAN;
    o.snr_db             = (float)(10 * log10((double)o.snr));
    o.noise_estimate_dbm = (float)(10 * log10((double)o.noise_estimate) + 30);
    o.cfo                = 0.f;
    res[row]             = o;
  }
}

} // namespace

struct srslte_hip_chest_ul {
  uint32_t cell_id, nof_prb, nsl; // nsl: symbols per slot (7, or 6 with the extended CP)
  srslte_hip_dmrs_pusch_cfg_t cfg;
  uint32_t n_prs[30][20], f_gh[20], v[20][30];
  // device DMRS of the grant last used: [10][2][12 * L_prb]
  cf32*    d_r;
  uint32_t r_L, r_n_dmrs;
  // per-PUSCH grants: every (L_prb, n_dmrs) seen so far keeps its table (srslte_chest_ul_pregen holds all of them at once, refsignal_ul.c:420-457)
  std::map<std::pair<uint32_t, uint32_t>, cf32*>* tables;
};

extern "C" srslte_hip_chest_ul_t* srslte_hip_chest_ul_create(uint32_t cell_id, uint32_t nof_prb, int cp_is_norm, const srslte_hip_dmrs_pusch_cfg_t* cfg)
{ // srslte_chest_ul_init + srslte_chest_ul_set_cell (chest_ul.c:51-194, refsignal_ul.c:206-238) + srslte_chest_ul_pregen
  if (cell_id > 503 || nof_prb < 6 || nof_prb > 110 || !cfg || cfg->cyclic_shift >= 8 || cfg->delta_ss >= 30) {
    hip_log("[srslte_hip] chest_ul: unsupported cell / DMRS configuration (id=%u prb=%u cp_norm=%d)\n", cell_id, nof_prb, cp_is_norm);
    return nullptr;
  }
  auto* q    = new srslte_hip_chest_ul();
  q->cell_id = cell_id;
  q->nof_prb = nof_prb;
  q->nsl     = cp_is_norm ? 7 : 6;
  q->cfg     = *cfg;
  q->d_r     = nullptr;
  q->r_L = q->r_n_dmrs = 0xffffffffu;
  q->tables = new std::map<std::pair<uint32_t, uint32_t>, cf32*>();
  std::vector<uint8_t> c;
  for (uint32_t ds = 0; ds < 30; ds++) { // generate_n_prs :118-141 and generate_srslte_sequence_hopping_v :149-163 share the seed
    gold(((cell_id / 30) << 5) + (((cell_id % 30) + ds) % 30), 8 * q->nsl * 20, c); // 8 bits per SC-FDMA symbol: the CP sets the stride
    for (uint32_t ns = 0; ns < 20; ns++) {
      uint32_t n = 0;
      for (int i = 0; i < 8; i++) n += (uint32_t)c[8 * q->nsl * ns + i] << i;
      q->n_prs[ds][ns] = n;
      q->v[ns][ds]     = c[ns];
    }
  }
  gold(cell_id / 30, 160, c); // srslte_group_hopping_f_gh, phy_common.c:419-436
  for (uint32_t ns = 0; ns < 20; ns++) {
    q->f_gh[ns] = 0;
    for (int i = 0; i < 8; i++) q->f_gh[ns] += (uint32_t)c[8 * ns + i] << i;
  }
  return q;
}

extern "C" void srslte_hip_chest_ul_destroy(srslte_hip_chest_ul_t* q)
{
  if (!q) return;
  if (q->d_r) (void)hipFree(q->d_r);
  for (auto& kv : *q->tables) (void)hipFree(kv.second);
  delete q->tables;
  delete q;
}

// srslte_refsignal_dmrs_pusch_gen (refsignal_ul.c:459-487): r_host [2][12 * L_prb]. The float / double mix of the reference is kept
// operation by operation (at 100 PRB the exponent's argument reaches 4e6 rad, where a float resolves 0.5 rad), including the
// fused multiply-add its -Ofast -mfma build makes of tmp_arg[i] + alpha * i.
extern "C" int srslte_hip_refsignal_dmrs_pusch_gen(const srslte_hip_chest_ul_t* q, uint32_t L_prb, uint32_t sf_idx, uint32_t n_dmrs, void* r_host)
{
  if (!q || !r_host || n_dmrs >= 8 || sf_idx >= 10 || L_prb > q->nof_prb) return SRSLTE_ERROR_INVALID_INPUTS;
  if (L_prb == 0) return SRSLTE_ERROR_INVALID_INPUTS;
  cf32*          r    = (cf32*)r_host;
  const uint32_t M_sc = 12 * L_prb, N_sz = L_prb >= 3 ? largest_prime_below(M_sc) : 1;
  for (uint32_t ns = 2 * sf_idx; ns < 2 * (sf_idx + 1); ns++) {
    const uint32_t u = ((q->cfg.group_hopping_en ? q->f_gh[ns] : 0) + (q->cell_id % 30) + q->cfg.delta_ss) % 30;
    const uint32_t v = (L_prb >= 6 && q->cfg.sequence_hopping_en) ? q->v[ns][q->cfg.delta_ss] : 0;
    const float    n_sz = (float)N_sz, q_hat = n_sz * (u + 1) / 31;
    float          qf;
    if ((((uint32_t)(2 * q_hat)) % 2) == 0) { // get_q :257-269
      qf = (float)(q_hat + 0.5 + v);
    } else {
      qf = (float)(q_hat + 0.5 - v);
    }
    const float    qq    = (float)(uint32_t)qf;
    const uint32_t n_cs  = (N_DMRS_1[q->cfg.cyclic_shift] + N_DMRS_2[n_dmrs] + q->n_prs[q->cfg.delta_ss][ns]) % 12; // pusch_alpha :296-304
    const float    alpha = (float)(2 * M_PI * n_cs / 12);
    for (uint32_t i = 0; i < M_sc; i++) {
      const float m   = (float)(i % N_sz);
      float       arg = (float)(-M_PI * qq * m * (m + 1) / n_sz); // arg_r_uv_mprb :271-283
      if (L_prb < 3) arg = (float)((2 * ((L_prb == 1 ? PHI_12 : PHI_24)[u][i] - '0') - 3) * M_PI / 4); // :143-147,:251-255
      const float x   = fmaf(alpha, (float)i, arg);
      r[(ns % 2) * M_sc + i] = make_float2(cosf(x), sinf(x));
    }
  }
  return SRSLTE_SUCCESS;
}

// Device DMRS of a grant, [10][2][12 * L_prb] (what srslte_chest_ul_pregen keeps for every (n_dmrs, sf, L): built per grant here)
int chest_ul_dmrs_table(srslte_hip_chest_ul_t* q, uint32_t L_prb, uint32_t n_dmrs, const void** d_r)
{
  if (q->r_L != L_prb || q->r_n_dmrs != n_dmrs) {
    std::vector<cf32> r((size_t)10 * 2 * 12 * L_prb);
    for (uint32_t sf = 0; sf < 10; sf++) {
      int rc = srslte_hip_refsignal_dmrs_pusch_gen(q, L_prb, sf, n_dmrs, r.data() + (size_t)sf * 2 * 12 * L_prb);
      if (rc) return rc;
    }
    if (q->d_r) (void)hipFree(q->d_r);
    q->d_r = nullptr;
    HIP_TRY(hipMalloc((void**)&q->d_r, sizeof(cf32) * r.size()));
    HIP_TRY(hipMemcpy(q->d_r, r.data(), sizeof(cf32) * r.size(), hipMemcpyHostToDevice));
    q->r_L      = L_prb;
    q->r_n_dmrs = n_dmrs;
  }
  *d_r = q->d_r;
  return SRSLTE_SUCCESS;
}

// The DMRS table of (L_prb, n_dmrs), [10][2][12 * L_prb], from the per-object cache of the grants modes (made on first use, kept)
int chest_ul_dmrs_table_cached(srslte_hip_chest_ul_t* q, uint32_t L_prb, uint32_t n_dmrs, const void** d_r)
{
  if (!q || !d_r || L_prb == 0 || L_prb > q->nof_prb || n_dmrs >= 8) return SRSLTE_ERROR_INVALID_INPUTS;
  auto it = q->tables->find({L_prb, n_dmrs});
  if (it == q->tables->end()) {
    std::vector<cf32> r((size_t)10 * 2 * 12 * L_prb);
    for (uint32_t sf = 0; sf < 10; sf++) {
      if (int rc = srslte_hip_refsignal_dmrs_pusch_gen(q, L_prb, sf, n_dmrs, r.data() + (size_t)sf * 2 * 12 * L_prb)) return rc;
    }
    cf32* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, sizeof(cf32) * r.size()));
    HIP_TRY(hipMemcpy(d, r.data(), sizeof(cf32) * r.size(), hipMemcpyHostToDevice));
    it = q->tables->emplace(std::make_pair(L_prb, n_dmrs), d).first;
  }
  *d_r = it->second;
  return SRSLTE_SUCCESS;
}

// Per-PUSCH grants: n_items PUSCHs of one (L_prb, n_dmrs) - d_items[i] names the subframe of the batch, the PRB offset of each slot and the row
// of d_res of PUSCH i - in one launch. The table of (L_prb, n_dmrs) is made on first use and kept.
int chest_ul_estimate_items(srslte_hip_chest_ul_t* q, uint32_t tti0, uint32_t L_prb, uint32_t n_dmrs, const ChestUlItem* d_items, int n_items,
                            const void* d_grid, void* d_ce, void* d_res, hipStream_t st)
{
  if (!q || !d_grid || !d_items || n_items < 0 || L_prb == 0 || L_prb > q->nof_prb || n_dmrs >= 8) return SRSLTE_ERROR_INVALID_INPUTS;
  if (n_items == 0) return SRSLTE_SUCCESS;
  const void* d_tab = nullptr;
  if (int rc = chest_ul_dmrs_table_cached(q, L_prb, n_dmrs, &d_tab)) return rc;
  ChestUlGeom g;
  g.cell_nre = 12 * (int)q->nof_prb; g.L_prb = (int)L_prb; g.n_prb = 0; g.n_prb1 = 0; g.tti0 = (int)tti0; g.w = 0.3333f; g.nsl = (int)q->nsl;
  hipLaunchKernelGGL(chest_ul_kernel, dim3(n_items), dim3(CH_THREADS), sizeof(cf32) * 2 * 12 * L_prb, st, (const cf32*)d_grid, (cf32*)d_ce,
                     (ChestUlResDev*)d_res, (const cf32*)d_tab, g, d_items);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}

// d_grid: [nof_sf][14 (12: extended CP)][12 * cell nof_prb]; d_ce: same shape (only the granted PRBs are written, as upstream) or NULL;
// d_res: [nof_sf] srslte_hip_chest_ul_res_t or NULL. Same grant (L_prb, n_prb in both slots, n_dmrs) for every subframe of the batch.
extern "C" int srslte_hip_chest_ul_estimate_pusch_batch(srslte_hip_chest_ul_t* q, uint32_t tti0, uint32_t L_prb, uint32_t n_prb, uint32_t n_dmrs,
                                                        const void* d_grid, void* d_ce, void* d_res, int nof_sf, void* stream)
{
  return srslte_hip_chest_ul_estimate_pusch_batch_hop(q, tti0, L_prb, n_prb, n_prb, n_dmrs, d_grid, d_ce, d_res, nof_sf, stream);
}

// The same with a PRB offset per slot (srslte_pusch_grant_t.n_prb[0 / 1]): intra-subframe hopping. The reference estimates and fills each slot at
// its own offset (chest_ul.c:244-266, no interpolation between the slots) and only prints a complaint (:293-295).
extern "C" int srslte_hip_chest_ul_estimate_pusch_batch_hop(srslte_hip_chest_ul_t* q, uint32_t tti0, uint32_t L_prb, uint32_t n_prb, uint32_t n_prb_slot1,
                                                            uint32_t n_dmrs, const void* d_grid, void* d_ce, void* d_res, int nof_sf, void* stream)
{
  if (!q || !d_grid || nof_sf < 0 || n_prb + L_prb > q->nof_prb || n_prb_slot1 + L_prb > q->nof_prb || n_dmrs >= 8) return SRSLTE_ERROR_INVALID_INPUTS;
  if (!srslte_hip_dft_precoding_valid_prb(L_prb)) {
    hip_log("[srslte_hip] Error invalid nof_prb=%u\n", L_prb); // chest_ul.c:278-281
    return SRSLTE_ERROR_INVALID_INPUTS;
  }
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  const void* d_r = nullptr;
  if (int rc = chest_ul_dmrs_table(q, L_prb, n_dmrs, &d_r)) return rc;
  ChestUlGeom g;
  g.cell_nre = 12 * (int)q->nof_prb; g.L_prb = (int)L_prb; g.n_prb = (int)n_prb; g.n_prb1 = (int)n_prb_slot1; g.tti0 = (int)tti0; g.w = 0.3333f; g.nsl = (int)q->nsl;
  hipLaunchKernelGGL(chest_ul_kernel, dim3(nof_sf), dim3(CH_THREADS), sizeof(cf32) * 2 * 12 * L_prb, (hipStream_t)stream, (const cf32*)d_grid,
                     (cf32*)d_ce, (ChestUlResDev*)d_res, (const cf32*)q->d_r, g, (const ChestUlItem*)nullptr);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// chest_common.c's two array helpers as stand-alone launches (the estimator kernels above do the same work inline): the single-call
// srslte_chest_average_pilots / srslte_chest_estimate_noise_pilots of the compatibility API (compat_refsignal.cpp) run these.
namespace {

__global__ void __launch_bounds__(CH_THREADS) chest_average_pilots_kernel(const cf32* __restrict__ in, cf32* __restrict__ out,
                                                                           const float* __restrict__ filt, int nof_ref, int filter_len)
{ // chest_common.c:95-101: one "same" convolution per symbol row (blockIdx.y)
  const int   l = blockIdx.y, i = blockIdx.x * CH_THREADS + threadIdx.x;
  if (i < nof_ref) out[(size_t)l * nof_ref + i] = conv_at(in + (size_t)l * nof_ref, filt, nof_ref, filter_len, i);
}

__global__ void __launch_bounds__(CH_THREADS) chest_noise_pilots_kernel(const cf32* __restrict__ noisy, const cf32* __restrict__ noiseless,
                                                                         cf32* __restrict__ noise_vec, int n, float* __restrict__ power)
{ // chest_common.c:51-60: noise_vec = noiseless - noisy, mean |noise_vec|^2
  __shared__ float red[CH_THREADS];
  float            acc = 0.f;
  for (int i = threadIdx.x; i < n; i += CH_THREADS) {
    const cf32 d = c_sub(noiseless[i], noisy[i]);
    noise_vec[i] = d;
    acc += d.x * d.x + d.y * d.y;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = CH_THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *power = red[0] / (float)n;
}

} // namespace

int chest_average_pilots_launch(const void* d_in, void* d_out, const float* d_filt, int nof_ref, int nof_symbols, int filter_len, hipStream_t st)
{
  if (nof_ref <= 0 || nof_symbols <= 0) return SRSLTE_SUCCESS;
  hipLaunchKernelGGL(chest_average_pilots_kernel, dim3((nof_ref + CH_THREADS - 1) / CH_THREADS, nof_symbols), dim3(CH_THREADS), 0, st,
                     (const cf32*)d_in, (cf32*)d_out, d_filt, nof_ref, filter_len);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}

int chest_noise_pilots_launch(const void* d_noisy, const void* d_noiseless, void* d_noise_vec, int n, float* d_power, hipStream_t st)
{
  hipLaunchKernelGGL(chest_noise_pilots_kernel, dim3(1), dim3(CH_THREADS), 0, st, (const cf32*)d_noisy, (const cf32*)d_noiseless,
                     (cf32*)d_noise_vec, n, d_power);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}
