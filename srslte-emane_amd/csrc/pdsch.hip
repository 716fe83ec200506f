// PDSCH / PUSCH pipelines for gfx950: keep IQ -> grid -> estimates -> LLRs -> soft buffers -> transport blocks (and the transmit
// directions) on the device for a whole batch of subframes (SURVEY §8b "batch entry points", §8f N1/N3/N4 glue).
//
// PDSCH receive stages and the reference code they replace:
//   ofdm_rx_kernel (fft.hip)          srslte_ofdm_rx_sf                                    ofdm.c:453-467
//   chest_dl_kernel (chest.hip)       srslte_chest_dl_estimate_cfg, 1/2/4 ports x 1-4 antennas  chest_dl.c:884-908
//   pdsch_demod_kernel (here)         srslte_pdsch_get + srslte_predecoding_single[_multi] (csi variants) + srslte_demod_soft_demodulate_s/_b
//                                     + srslte_scrambling_s/sb_offset                      pdsch.c:81-206,:760-779,:890-935, precoding.c:251-348
//   pdsch_demod_div[4]_kernel (here)  the same with srslte_predecoding_diversity_csi + srslte_layerdemap_diversity (TM2)  precoding.c:564-650
//   rm_rx[_lds]_kernel (here)         srslte_rm_turbo_rx_lut[_8bit] per code block, HARQ combining, csi_correction  sch.c:318-346, rm_turbo.c:374-465,
//                                                                                          pdsch.c:574-690
//   tdec_*_kernel (tdec.hip)          srslte_tdec_new_cb/_iteration[_8bit] + CB CRC, skip of blocks decoded earlier  sch.c:317-383
//   tb_asm_kernel / tb_crc_kernel     payload assembly + TB CRC24A                         sch.c:401-410,:470-488
// One codeword, TM1 or transmit diversity, full-band grant, FDD, normal CP. Further down: the PUSCH receive pipeline (eNB), the PUSCH
// transmit pipeline (UE) and the PDSCH transmit pipeline (eNB), each with its own header comment.
//
// The fused demapper kernels gather each PDSCH RE and its channel estimates once, equalise with exact divisions (the reference's
// single-port csi variant multiplies by the 12-bit _mm256_rcp_ps approximation, precoding.c:262-275, whose value is CPU-vendor
// dependent; LLRs may therefore differ from a given host's by an LSB, decoded blocks do not), demap, descramble and write the LLRs
// through LDS with 16-byte stores: 16 B read and Qm * sizeof(LLR) B written per RE, no intermediate d/e round trips through HBM.
#include "common.hpp"
#include "demod_dev.hpp"
#include "phy_hip_internal.hpp"
#include <algorithm>
#include <map>
#include <math.h>
#include <string.h>
#include <vector>

namespace {

struct ChestResDev {
  float noise_estimate, noise_estimate_dbm, snr_db, rsrp, rsrp_dbm, rsrq, rsrq_db, rssi_dbm, cfo, sync_error;
};

struct SfClass { // RE list per subframe class: 0 = sf 0 (PSS/SSS+PBCH), 1 = sf 5 (PSS/SSS), 2 = the rest
  const uint32_t* idx;
  int             nof_re;
};

// Per-subframe / per-code-block descriptors of srslte_hip_dl_rx_batch_grants: every subframe of a batch carries its own grant
// (srslte_pdsch_grant_t: PRB masks of both slots, modulation, transport block size, redundancy version; RNTI, CFI). Null descriptor
// pointers in the geometry structs = the one fixed full-band configuration of srslte_hip_dl_rx_batch.
struct SfDesc {
  const uint32_t* idx; // RE list of the subframe (pdsch_relist_kernel)
  const uint32_t* scr; // packed scrambling bits of the subframe (scr_gen_kernel)
  int             nof_re, mod, Qm;
  int             C, K, tbs, rlen; // segmentation of its transport block (36.212 5.1.2): C blocks of K bits, rlen payload bits per block
  const uint32_t* crc_fac;         // [256] x^(8 cB (255 - t)) mod g_CRC24A, cB = ceil((tbs / 8 + 3) / 256): tb_crc_bytes_kernel's chunk weights
  int             scheme, codebook, nof_tb; // srslte_tx_scheme_t of the subframe's grant (0 / 1: the cell's single-port or diversity mode; 2, 3: two-layer
                                            // modes), pre-decoder codebook index, transport blocks; codeword 1 has its own entry max_batch further on
};
struct CbDesc {
  int             sf, cb;  // subframe of the batch, code block of its transport block
  int             C, K, Qm, nof_re;
  int             combine; // 0: new data, the soft buffer is overwritten; 1: retransmission, added (and skipped if the block's CRC passed)
  int             w_len;   // soft-buffer slots of this block length (multiple of 32) = stride of its slot table
  const uint32_t* tbl;     // slot table of (K, rv)
  int             Nl;      // the block split counts in units of Qm * Nl bits: 2 for transmit diversity, else 1 (sch.c:507-531)
  int             e_off;   // LLRs in front of the shared channel's in this block's row (PUSCH: the CQI report's)
};
struct GrantDev { // what the list / sequence kernels need of a grant
  uint32_t mask[2][4]; // prb_idx[s][n] as bits
  int      sf_idx, lstart, q_off, rnti;
  int      cw;         // codeword of the scrambling sequence (36.211 6.3.1: q << 13 in c_init)
};

struct PdschGeom {
  const SfDesc* desc; // grants mode, else null
  SfClass cls[3];
  int     grid_len;   // 14 * 12 * nof_prb
  int     max_re, max_bits, mod, Qm, mmse, scr_words, tti0, nof_rx, nof_ports;
  float     inv_scaling; // 1 / pdsch_scaling (pdsch.c:852-858): 1, or 1 / rho_a with cfg.power_scale
  float*    csi;     // [nof_sf][max_re] channel gain per RE, or null (cfg.csi_enable)
  uint32_t* csi_max; // [nof_sf] bit pattern of the largest gain of each subframe (non-negative floats order like their bits), zeroed per call
  // two-layer modes (pdsch_demod_mimo_kernel): srslte_tx_scheme_t, codebook index, and the second codeword's modulation and buffers
  int             tx_scheme, codebook_idx, nof_tb, mod1, Qm1, max_bits1, scr_words1;
  const uint32_t* scr1;
  float*          csi1;
  uint32_t*       csi_max1;
  int             cw1_off; // grants mode: codeword 1 of subframe sf is described by desc[cw1_off + sf]
};

// the subframe's largest csi: wavefront maximum, one atomic per wavefront
__device__ __forceinline__ void csi_note_max(uint32_t* dst, float v)
{
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(dst, __float_as_uint(v));
}

__device__ __forceinline__ int sf_class(int sf_idx) { return sf_idx == 0 ? 0 : (sf_idx == 5 ? 1 : 2); }

// grid = (ceil(max_re/256), nof_sf). LLR = int16_t (srslte_demod_soft_demodulate_s + srslte_scrambling_s_offset) or int8_t
// (srslte_demod_soft_demodulate_b + srslte_scrambling_sb_offset, the q->llr_is_8bit branch of pdsch.c:760-779)
template <typename LLR>
__global__ __launch_bounds__(256) void pdsch_demod_kernel(const cf32* __restrict__ grid, const cf32* __restrict__ ce,
                                                          const ChestResDev* __restrict__ res, const uint32_t* __restrict__ scr,
                                                          cf32* __restrict__ d_out, LLR* __restrict__ e_out, PdschGeom g)
{
  __shared__ __attribute__((aligned(16))) LLR stage[256 * 8]; // the workgroup's LLRs, written out with 16-byte stores
  const int       sf = blockIdx.y, sf_idx = (g.tti0 + sf) % 10;
  SfClass         c  = g.cls[sf_class(sf_idx)];
  int             mod = g.mod, Qm = g.Qm;
  const uint32_t* cs = scr + (size_t)sf_idx * g.scr_words; // one spare word behind every sequence
  if (g.desc) {
    const SfDesc d = g.desc[sf];
    if (d.scheme >= 2) return; // a two-layer grant: pdsch_demod_mimo_kernel's
    c.idx = d.idx; c.nof_re = d.nof_re; mod = d.mod; Qm = d.Qm; cs = d.scr;
  }
  const int     base = blockIdx.x * blockDim.x, i = base + threadIdx.x;
  if (base >= c.nof_re) return;
  const bool     live = i < c.nof_re;
  const uint32_t k  = c.idx[live ? i : c.nof_re - 1];
  const float    n0 = g.mmse ? res[sf].noise_estimate : 0.f;
  cf32           x;
  float          gain; // srslte_predecoding_single_csi's side output (precoding.c:251-291)
  if (g.nof_rx == 1) {
    const cf32 y = grid[(size_t)sf * g.grid_len + k], h = ce[(size_t)sf * g.grid_len + k];
    // precoding.c:277-288 with scaling = 1 (pdsch.c:852-858, power_scale off)
    const float re = y.x * h.x + y.y * h.y, im = y.y * h.x - y.x * h.y, csi = h.x * h.x + h.y * h.y + n0;
    x = make_float2(re * g.inv_scaling / csi, im * g.inv_scaling / csi);
    gain = csi;
  } else { // srslte_predecoding_single_multi (precoding.c:138-262): maximum-ratio combining over the receive antennas
    float re = 0.f, im = 0.f, hh = 0.f;
    for (int a = 0; a < g.nof_rx; a++) {
      const size_t o = ((size_t)sf * g.nof_rx + a) * g.grid_len + k;
      const cf32   y = grid[o], h = ce[o];
      const float  pr = y.x * h.x + y.y * h.y, pi = y.y * h.x - y.x * h.y, ph = h.x * h.x + h.y * h.y;
      re = a ? re + pr : pr;
      im = a ? im + pi : pi;
      hh = a ? hh + ph : ph;
    }
    if (n0 > 0.f) hh += n0;
    x = make_float2(re / hh * g.inv_scaling, im / hh * g.inv_scaling);
    gain = hh;
  }
  if (g.csi) {
    if (live) g.csi[(size_t)sf * g.max_re + i] = gain;
    csi_note_max(g.csi_max + sf, live ? gain : 0.f);
  }
  if (d_out && live) d_out[(size_t)sf * g.max_re + i] = x;
  LLR o[8];
  if constexpr (sizeof(LLR) == 1) {
    demod_dev::demod_b(mod, x, i, c.nof_re, o);
  } else {
    demod_dev::demod_s(mod, x, i, c.nof_re, o);
  }
  const int       bit0 = (live ? i : 0) * Qm;
  const uint32_t  c2   = (uint32_t)((((uint64_t)cs[(bit0 >> 5) + 1] << 32) | cs[bit0 >> 5]) >> (bit0 & 31));
  for (int j = 0; j < Qm; j++) {
    LLR v = o[j];
    if ((c2 >> j) & 1) v = (LLR)-v; // scrambling.c:45-51: sign instruction, -(-min) stays min
    stage[threadIdx.x * Qm + j] = v;
  }
  __syncthreads();
  // max_bits is a multiple of 16 and so is 256 * Qm: the workgroup's output starts on a 16-byte boundary
  const int   nbytes = min(256, c.nof_re - base) * Qm * (int)sizeof(LLR);
  char*       dst    = reinterpret_cast<char*>(e_out + (size_t)sf * g.max_bits + (size_t)base * Qm);
  const char* src    = reinterpret_cast<const char*>(stage);
  for (int o16 = threadIdx.x * 16; o16 + 16 <= nbytes; o16 += 256 * 16) *reinterpret_cast<uint4*>(dst + o16) = *reinterpret_cast<const uint4*>(src + o16);
  const int rem = nbytes & 15;
  if ((int)threadIdx.x < rem) dst[nbytes - rem + threadIdx.x] = src[nbytes - rem + threadIdx.x];
}

// 2-port transmit diversity (TM2): srslte_predecoding_diversity_csi for 2 ports + srslte_layerdemap_diversity (precoding.c:564-598,
// layermap.c:140-148; pdsch.c:890-935 with tx_scheme DIVERSITY) fused with the demapper and descrambler. One thread per PAIR of
// consecutive PDSCH REs (2i, 2i+1) = one SFBC block: x0 = sum_a h00* r0 + h11 r1*, x1 = sum_a -h10 r0* + h01* r1, both divided by
// sum_a |h00|^2 + |h11|^2 and scaled by sqrt(2); d[2i] = x0, d[2i+1] = x1. ce is [sf][port][antenna][grid]. grid = (ceil(max_re/512), nof_sf).
template <typename LLR>
__global__ __launch_bounds__(256) void pdsch_demod_div_kernel(const cf32* __restrict__ grid, const cf32* __restrict__ ce,
                                                              const uint32_t* __restrict__ scr, cf32* __restrict__ d_out, LLR* __restrict__ e_out,
                                                              PdschGeom g)
{
  __shared__ __attribute__((aligned(16))) LLR stage[512 * 8];
  const int       sf = blockIdx.y, sf_idx = (g.tti0 + sf) % 10;
  SfClass         c  = g.cls[sf_class(sf_idx)];
  const uint32_t* cs = scr + (size_t)sf_idx * g.scr_words; // one spare word behind every sequence
  int             mod = g.mod, Qm = g.Qm; // locals: writing to the by-value argument would move it to scratch
  if (g.desc) { // per-subframe grants
    const SfDesc d = g.desc[sf];
    if (d.scheme >= 2) return; // a two-layer grant: pdsch_demod_mimo_kernel's
    c.idx = d.idx; c.nof_re = d.nof_re; mod = d.mod; Qm = d.Qm; cs = d.scr;
  }
  const int     base = blockIdx.x * 512, i0 = base + 2 * threadIdx.x; // nof_re is even for a 2-port cell
  if (base >= c.nof_re) return;
  const bool     live = i0 < c.nof_re;
  const uint32_t k0 = c.idx[live ? i0 : c.nof_re - 2], k1 = c.idx[live ? i0 + 1 : c.nof_re - 1];
  float          hh = 0.f, x0r = 0.f, x0i = 0.f, x1r = 0.f, x1i = 0.f;
  for (int a = 0; a < g.nof_rx; a++) {
    const cf32* y  = grid + ((size_t)sf * g.nof_rx + a) * g.grid_len;
    const cf32* h0 = ce + (((size_t)sf * 2 + 0) * g.nof_rx + a) * g.grid_len;
    const cf32* h1 = ce + (((size_t)sf * 2 + 1) * g.nof_rx + a) * g.grid_len;
    const cf32  r0 = y[k0], r1 = y[k1], h00 = h0[k0], h01 = h0[k1], h10 = h1[k0], h11 = h1[k1];
    hh += h00.x * h00.x + h00.y * h00.y + h11.x * h11.x + h11.y * h11.y;
    if (hh == 0.f) hh = 1e-4f;
    x0r += h00.x * r0.x + h00.y * r0.y + h11.x * r1.x + h11.y * r1.y;
    x0i += h00.x * r0.y - h00.y * r0.x + h11.y * r1.x - h11.x * r1.y;
    x1r += -(h10.x * r0.x + h10.y * r0.y) + h01.x * r1.x + h01.y * r1.y;
    x1i += -(h10.y * r0.x - h10.x * r0.y) + h01.x * r1.y - h01.y * r1.x;
  }
  if (g.csi) { // csi[2i] = csi[2i + 1] = hh (precoding.c:590-591)
    if (live) *reinterpret_cast<float2*>(g.csi + (size_t)sf * g.max_re + i0) = make_float2(hh, hh);
    csi_note_max(g.csi_max + sf, live ? hh : 0.f);
  }
  hh *= 1.0f / g.inv_scaling; // hh *= scaling (precoding.c:593)
  const cf32 x[2] = {make_float2((float)((double)(x0r / hh) * 1.4142135623730951), (float)((double)(x0i / hh) * 1.4142135623730951)),
                     make_float2((float)((double)(x1r / hh) * 1.4142135623730951), (float)((double)(x1i / hh) * 1.4142135623730951))};
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const int i = (live ? i0 : 0) + t;
    if (d_out && live) d_out[(size_t)sf * g.max_re + i] = x[t];
    LLR o[8];
    if constexpr (sizeof(LLR) == 1) {
      demod_dev::demod_b(mod, x[t], i, c.nof_re, o);
    } else {
      demod_dev::demod_s(mod, x[t], i, c.nof_re, o);
    }
    const int      bit0 = i * Qm;
    const uint32_t c2   = (uint32_t)((((uint64_t)cs[(bit0 >> 5) + 1] << 32) | cs[bit0 >> 5]) >> (bit0 & 31));
    for (int j = 0; j < Qm; j++) stage[(2 * threadIdx.x + t) * Qm + j] = ((c2 >> j) & 1) ? (LLR)-o[j] : o[j];
  }
  __syncthreads();
  const int   nbytes = min(512, c.nof_re - base) * Qm * (int)sizeof(LLR);
  char*       dst    = reinterpret_cast<char*>(e_out + (size_t)sf * g.max_bits + (size_t)base * Qm);
  const char* src    = reinterpret_cast<const char*>(stage);
  for (int o16 = threadIdx.x * 16; o16 + 16 <= nbytes; o16 += 256 * 16) *reinterpret_cast<uint4*>(dst + o16) = *reinterpret_cast<const uint4*>(src + o16);
  const int rem = nbytes & 15;
  if ((int)threadIdx.x < rem) dst[nbytes - rem + threadIdx.x] = src[nbytes - rem + threadIdx.x];
}

// 4-port transmit diversity (SFBC + FSTD): srslte_predecoding_diversity_csi for 4 ports + srslte_layerdemap_diversity (precoding.c:599-650,
// layermap.c:140-148). One thread per group of four consecutive PDSCH REs: sub-carriers 4i, 4i+1 carry the Alamouti pair of ports 0/2,
// 4i+2, 4i+3 that of ports 1/3; every symbol has its own divisor. grid = (ceil(max_re/1024), nof_sf).
template <typename LLR>
__global__ __launch_bounds__(256) void pdsch_demod_div4_kernel(const cf32* __restrict__ grid, const cf32* __restrict__ ce,
                                                               const uint32_t* __restrict__ scr, cf32* __restrict__ d_out, LLR* __restrict__ e_out,
                                                               PdschGeom g)
{
  __shared__ __attribute__((aligned(16))) LLR stage[1024 * 8];
  const int       sf = blockIdx.y, sf_idx = (g.tti0 + sf) % 10;
  SfClass         c  = g.cls[sf_class(sf_idx)];
  const uint32_t* cs = scr + (size_t)sf_idx * g.scr_words;
  int             mod = g.mod, Qm = g.Qm; // locals: writing to the by-value argument would move it to scratch
  if (g.desc) { // per-subframe grants
    const SfDesc d = g.desc[sf];
    if (d.scheme >= 2) return;
    c.idx = d.idx; c.nof_re = d.nof_re; mod = d.mod; Qm = d.Qm; cs = d.scr;
  }
  const int     base = blockIdx.x * 1024, i0 = base + 4 * threadIdx.x; // nof_re is a multiple of 4 for a 4-port cell
  if (base >= c.nof_re) return;
  const bool live = i0 < c.nof_re;
  uint32_t   k[4];
#pragma unroll
  for (int t = 0; t < 4; t++) k[t] = c.idx[live ? i0 + t : c.nof_re - 4 + t];
  float a[4] = {0.f, 0.f, 0.f, 0.f}, xr[4] = {0.f, 0.f, 0.f, 0.f}, xi[4] = {0.f, 0.f, 0.f, 0.f};
  for (int an = 0; an < g.nof_rx; an++) {
    const cf32* y = grid + ((size_t)sf * g.nof_rx + an) * g.grid_len;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const cf32* hA = ce + (((size_t)sf * 4 + half) * g.nof_rx + an) * g.grid_len;     // port 0 / 1
      const cf32* hB = ce + (((size_t)sf * 4 + 2 + half) * g.nof_rx + an) * g.grid_len; // port 2 / 3
      const uint32_t k0 = k[2 * half], k1 = k[2 * half + 1];
      const cf32     h00 = hA[k0], h01 = hB[k0], h10 = hA[k1], h11 = hB[k1], r0 = y[k0], r1 = y[k1];
      a[2 * half] += h00.x * h00.x + h00.y * h00.y + h11.x * h11.x + h11.y * h11.y;
      a[2 * half + 1] += h10.x * h10.x + h10.y * h10.y + h01.x * h01.x + h01.y * h01.y;
      xr[2 * half] += h00.x * r0.x + h00.y * r0.y + h11.x * r1.x + h11.y * r1.y;
      xi[2 * half] += h00.x * r0.y - h00.y * r0.x + h11.y * r1.x - h11.x * r1.y;
      xr[2 * half + 1] += -(h01.x * r0.x + h01.y * r0.y) + h10.x * r1.x + h10.y * r1.y;
      xi[2 * half + 1] += -(h01.y * r0.x - h01.x * r0.y) + h10.x * r1.y - h10.y * r1.x;
    }
  }
  float           gmax = 0.f;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int   i  = (live ? i0 : 0) + t;
    const float at = a[t] * (1.0f / g.inv_scaling); // a *= scaling (precoding.c:634-637)
    const cf32  x  = make_float2(xr[t] / at * 1.41421356f, xi[t] / at * 1.41421356f);
    if (g.csi && live) g.csi[(size_t)sf * g.max_re + i] = at / g.nof_rx; // precoding.c:639-642
    gmax = fmaxf(gmax, live ? at / g.nof_rx : 0.f);
    if (d_out && live) d_out[(size_t)sf * g.max_re + i] = x;
    LLR o[8];
    if constexpr (sizeof(LLR) == 1) {
      demod_dev::demod_b(mod, x, i, c.nof_re, o);
    } else {
      demod_dev::demod_s(mod, x, i, c.nof_re, o);
    }
    const int      bit0 = i * Qm;
    const uint32_t c2   = (uint32_t)((((uint64_t)cs[(bit0 >> 5) + 1] << 32) | cs[bit0 >> 5]) >> (bit0 & 31));
    for (int j = 0; j < Qm; j++) stage[(4 * threadIdx.x + t) * Qm + j] = ((c2 >> j) & 1) ? (LLR)-o[j] : o[j];
  }
  if (g.csi) csi_note_max(g.csi_max + sf, gmax);
  __syncthreads();
  const int   nbytes = min(1024, c.nof_re - base) * Qm * (int)sizeof(LLR);
  char*       dst    = reinterpret_cast<char*>(e_out + (size_t)sf * g.max_bits + (size_t)base * Qm);
  const char* src    = reinterpret_cast<const char*>(stage);
  for (int o16 = threadIdx.x * 16; o16 + 16 <= nbytes; o16 += 256 * 16) *reinterpret_cast<uint4*>(dst + o16) = *reinterpret_cast<const uint4*>(src + o16);
  const int rem = nbytes & 15;
  if ((int)threadIdx.x < rem) dst[nbytes - rem + threadIdx.x] = src[nbytes - rem + threadIdx.x];
}

// Two-layer modes on a 2-port cell with 2 receive antennas: large-delay CDD (TM3; srslte_predecoding_ccd_2x2_mmse_csi,
// precoding.c:918-1014) and closed-loop multiplexing (TM4; srslte_predecoding_multiplex_2x2_mmse_csi :1326-1438, one layer:
// srslte_predecoding_multiplex_2x1_mrc_csi :1624-1707), each with srslte_mat_2x2_mmse_csi_gen's algebra (mat.c) in exact divisions, fused with
// the demapper and descrambler of BOTH codewords (nof_layers == nof_tb in every case ra_dl.c:556-600 lets through: layer l is codeword l,
// no layer de-mapping; srslte_pdsch_codeword_decode pdsch.c:729-790 with the codeword's own modulation and scrambling sequence).
// One thread per PDSCH RE; ce is [sf][port][antenna][grid], grid [sf][antenna][grid]. grid = (ceil(max_re/256), nof_sf).
__device__ __forceinline__ cf32 cmul(cf32 a, cf32 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cf32 cmulc(cf32 a, cf32 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); } // a conj(b)
__device__ __forceinline__ cf32 cadd(cf32 a, cf32 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf32 csub(cf32 a, cf32 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf32 cmulj(cf32 a) { return make_float2(-a.y, a.x); }

template <typename LLR>
__global__ __launch_bounds__(256) void pdsch_demod_mimo_kernel(const cf32* __restrict__ grid, const cf32* __restrict__ ce,
                                                               const ChestResDev* __restrict__ res, const uint32_t* __restrict__ scr,
                                                               cf32* __restrict__ d_out0, cf32* __restrict__ d_out1, LLR* __restrict__ e_out0,
                                                               LLR* __restrict__ e_out1, PdschGeom g)
{
  __shared__ __attribute__((aligned(16))) LLR stage[2][256 * 8];
  const int       sf = blockIdx.y, sf_idx = (g.tti0 + sf) % 10;
  SfClass         c  = g.cls[sf_class(sf_idx)];
  int             tx_scheme = g.tx_scheme, codebook_idx = g.codebook_idx, nof_tb = g.nof_tb, mods[2] = {g.mod, g.mod1}, Qms[2] = {g.Qm, g.Qm1};
  const uint32_t* css[2] = {scr + (size_t)sf_idx * g.scr_words, g.scr1 ? g.scr1 + (size_t)sf_idx * g.scr_words1 : nullptr};
  if (g.desc) { // per-subframe grants: this subframe's own scheme, allocation, modulations and sequences
    const SfDesc d0 = g.desc[sf];
    if (d0.scheme < 2) return; // single antenna port or transmit diversity: the other kernels'
    const SfDesc d1 = g.desc[g.cw1_off + sf];
    c.idx = d0.idx; c.nof_re = d0.nof_re; tx_scheme = d0.scheme; codebook_idx = d0.codebook; nof_tb = d0.nof_tb;
    mods[0] = d0.mod; Qms[0] = d0.Qm; css[0] = d0.scr;
    mods[1] = d1.mod; Qms[1] = d1.Qm; css[1] = d1.scr;
  }
  const int     base = blockIdx.x * blockDim.x, i = base + threadIdx.x;
  if (base >= c.nof_re) return;
  const bool     live = i < c.nof_re;
  const int      ii = live ? i : c.nof_re - 1;
  const uint32_t k  = c.idx[ii];
  const float    n0 = g.mmse ? res[sf].noise_estimate : 0.f;
  const cf32*    yb = grid + (size_t)sf * 2 * g.grid_len;
  const cf32*    hb = ce + (size_t)sf * 4 * g.grid_len;
  const cf32     y0 = yb[k], y1 = yb[g.grid_len + k];
  const cf32     p0a0 = hb[k], p0a1 = hb[(size_t)g.grid_len + k], p1a0 = hb[(size_t)2 * g.grid_len + k], p1a1 = hb[(size_t)3 * g.grid_len + k];
  const float    scaling = 1.0f / g.inv_scaling;
  cf32           x[2];
  float          csi[2];
  if (nof_tb == 1) { // one layer: the codebook column applied to the ports, maximum-ratio combining over the antennas
    cf32 h0, h1;
    switch (codebook_idx) {
      case 0: h0 = cadd(p0a0, p1a0); h1 = cadd(p0a1, p1a1); break;
      case 1: h0 = csub(p0a0, p1a0); h1 = csub(p0a1, p1a1); break;
      case 2: h0 = cadd(p0a0, cmulj(p1a0)); h1 = cadd(p0a1, cmulj(p1a1)); break;
      default: h0 = csub(p0a0, cmulj(p1a0)); h1 = csub(p0a1, cmulj(p1a1)); break;
    }
    const float norm = 1.41421356f / scaling;
    const float cs = h0.x * h0.x + h0.y * h0.y + h1.x * h1.x + h1.y * h1.y, hh = norm / cs;
    const cf32  t  = cadd(cmulc(y0, h0), cmulc(y1, h1)); // conj(h0) y0 + conj(h1) y1
    x[0]   = make_float2(t.x * hh, t.y * hh);
    x[1]   = x[0];
    csi[0] = cs / norm * 0.70710678f;
    csi[1] = 0.f;
  } else {
    cf32  h00, h01, h10, h11; // effective channel: h[antenna][layer]
    float norm = 2.0f / scaling;
    if (tx_scheme == 3) { // H W U D(i): the sign of the second port alternates with the symbol index
      if (!(ii & 1)) {
        h00 = cadd(p0a0, p1a0); h10 = cadd(p0a1, p1a1); h01 = csub(p0a0, p1a0); h11 = csub(p0a1, p1a1);
      } else {
        h00 = csub(p0a0, p1a0); h10 = csub(p0a1, p1a1); h01 = cadd(p0a0, p1a0); h11 = cadd(p0a1, p1a1);
      }
    } else if (codebook_idx == 0) {
      h00 = p0a0; h01 = p1a0; h10 = p0a1; h11 = p1a1;
      norm = 1.41421356f / scaling;
    } else if (codebook_idx == 1) {
      h00 = cadd(p0a0, p1a0); h01 = csub(p0a0, p1a0); h10 = cadd(p0a1, p1a1); h11 = csub(p0a1, p1a1);
    } else {
      h00 = cadd(p0a0, cmulj(p1a0)); h01 = csub(p0a0, cmulj(p1a0)); h10 = cadd(p0a1, cmulj(p1a1)); h11 = csub(p0a1, cmulj(p1a1));
    }
    // A = H'H + N0 I; B = norm A^-1; W = B H'; x = W y; csi_l = 1 / Re(B_ll)
    cf32 a00 = cadd(cmulc(h00, h00), cmulc(h10, h10)), a11 = cadd(cmulc(h01, h01), cmulc(h11, h11));
    a00.x += n0;
    a11.x += n0;
    const cf32  a01 = cadd(cmulc(h01, h00), cmulc(h11, h10)), a10 = cadd(cmulc(h00, h01), cmulc(h10, h11));
    const cf32  det = csub(cmul(a00, a11), cmul(a01, a10));
    const float dm  = det.x * det.x + det.y * det.y;
    const cf32  nr  = make_float2(norm * (det.x / dm), norm * (-det.y / dm));
    const cf32  b00 = cmul(a11, nr), b01 = cmul(make_float2(-a01.x, -a01.y), nr), b10 = cmul(make_float2(-a10.x, -a10.y), nr), b11 = cmul(a00, nr);
    const cf32  w00 = cadd(cmulc(b00, h00), cmulc(b01, h01)), w01 = cadd(cmulc(b00, h10), cmulc(b01, h11));
    const cf32  w10 = cadd(cmulc(b10, h00), cmulc(b11, h01)), w11 = cadd(cmulc(b10, h10), cmulc(b11, h11));
    x[0]   = cadd(cmul(y0, w00), cmul(y1, w01));
    x[1]   = cadd(cmul(y0, w10), cmul(y1, w11));
    csi[0] = 1.0f / b00.x;
    csi[1] = 1.0f / b11.x;
  }
#pragma unroll
  for (int cw = 0; cw < 2; cw++) {
    if (cw >= nof_tb) break;
    const int       mod = mods[cw], Qm = Qms[cw];
    float*          csi_o = cw ? g.csi1 : g.csi;
    cf32*           d_out = cw ? d_out1 : d_out0;
    if (csi_o) {
      if (live) csi_o[(size_t)sf * g.max_re + i] = csi[cw];
      csi_note_max((cw ? g.csi_max1 : g.csi_max) + sf, live ? csi[cw] : 0.f);
    }
    if (d_out && live) d_out[(size_t)sf * g.max_re + i] = x[cw];
    LLR o[8];
    if constexpr (sizeof(LLR) == 1) {
      demod_dev::demod_b(mod, x[cw], i, c.nof_re, o);
    } else {
      demod_dev::demod_s(mod, x[cw], i, c.nof_re, o);
    }
    const uint32_t* cs   = css[cw];
    const int       bit0 = (live ? i : 0) * Qm;
    const uint32_t  c2   = (uint32_t)((((uint64_t)cs[(bit0 >> 5) + 1] << 32) | cs[bit0 >> 5]) >> (bit0 & 31));
    for (int j = 0; j < Qm; j++) stage[cw][threadIdx.x * Qm + j] = ((c2 >> j) & 1) ? (LLR)-o[j] : o[j];
  }
  __syncthreads();
#pragma unroll
  for (int cw = 0; cw < 2; cw++) {
    if (cw >= nof_tb) break;
    const int   Qm = Qms[cw], nbytes = min(256, c.nof_re - base) * Qm * (int)sizeof(LLR);
    char*       dst = reinterpret_cast<char*>((cw ? e_out1 : e_out0) + (size_t)sf * (cw ? g.max_bits1 : g.max_bits) + (size_t)base * Qm);
    const char* src = reinterpret_cast<const char*>(stage[cw]);
    for (int o16 = threadIdx.x * 16; o16 + 16 <= nbytes; o16 += 256 * 16) *reinterpret_cast<uint4*>(dst + o16) = *reinterpret_cast<const uint4*>(src + o16);
    const int rem = nbytes & 15;
    if ((int)threadIdx.x < rem) dst[nbytes - rem + threadIdx.x] = src[nbytes - rem + threadIdx.x];
  }
}

struct RmGeom {
  const CbDesc* cbd;       // grants mode: one entry per launched block (then C = code-block slots per subframe), else null
  uint8_t*      cb_ok_rst; // grants mode: CRC flags, cleared here for blocks that start new data
  int C, K, Qm, tti0, max_bits, w_stride, out_len; // out_len = 3K+12
  int nof_re[3];
  int             max_re, mod;
  int             Nl;      // the code-block split counts in units of Qm * N_L bits, N_L = 2 for transmit diversity (sch.c:507-531)
  const float*    csi;     // as PdschGeom; null = no CSI weighting
  const uint32_t* csi_max;
  int             combine; // HARQ: add to the soft buffer kept from earlier transmissions (rm_turbo.c:407-409 accumulates) instead of writing it
  const uint8_t*  skip;    // HARQ: [B*C] blocks whose CRC already passed are not touched (sch.c:317-318)
  int             e_off;   // LLRs in front of the shared channel's in every subframe (PUSCH: the CQI report's, sch.c:1058-1064)
};

// wrapping lane-wise add of packed int16 / int8 (the soft buffer accumulates with plain C '+=' on int16_t / int8_t)
template <typename LLR>
__device__ __forceinline__ uint32_t add_wrap(uint32_t a, uint32_t b)
{
  if constexpr (sizeof(LLR) == 2) {
    return (((a & 0xffffu) + (b & 0xffffu)) & 0xffffu) | (((a >> 16) + (b >> 16)) << 16);
  } else {
    const uint32_t lo = ((a & 0x00ff00ffu) + (b & 0x00ff00ffu)) & 0x00ff00ffu, hi = ((a & 0xff00ff00u) >> 8) + ((b & 0xff00ff00u) >> 8);
    return lo | ((hi & 0x00ff00ffu) << 8);
  }
}

// csi_correction (pdsch.c:574-690) applied to LLR number b of a subframe as it is read: 16-bit LLRs: (e * w) >> 16 with w = the gain of
// "its" symbol scaled to 32767 at the subframe's maximum, rounded to nearest even and saturated, for the whole groups of 4 / 4 / 12 / 8
// LLRs the SSE loops cover, (int16)(e * gain / max) for the symbols behind them; in the two-symbol groups (QPSK, 64QAM) the reference's
// _mm_blend_ps takes the low lanes from the SECOND symbol: reproduced. 8-bit LLRs: (int8)(e * (gain / max)).
struct CsiW {
  const float* csi; // the subframe's gains
  float        scale, inv;
  int          Qm, mod, body_bits, nsym;
};
__device__ __forceinline__ CsiW csi_setup(const RmGeom& g, int sf, int nsym, int Qm)
{ // Qm: of this block's transport block (grants mode: from its descriptor); srslte_mod_t = Qm / 2
  CsiW c;
  c.csi = g.csi + (size_t)sf * g.max_re;
  const float mx = nsym > 0 ? __uint_as_float(g.csi_max[sf]) : 1.0f;
  c.scale = 32767.0f / mx;
  c.inv   = 1.0f / mx;
  c.Qm = Qm; c.mod = Qm / 2; c.nsym = nsym;
  const int G = c.mod == 3 ? 12 : (c.mod == 4 ? 8 : 4);
  c.body_bits = (nsym * Qm / G) * G;
  return c;
}
template <typename LLR>
__device__ __forceinline__ LLR csi_apply(const CsiW& c, LLR v, int b)
{
  if constexpr (sizeof(LLR) == 1) {
    const int s = min(b / c.Qm, c.nsym - 1);
    return (LLR)((float)v * (c.csi[s] * c.inv));
  } else {
    if (b >= c.body_bits) {
      const int s = min(b / c.Qm, c.nsym - 1);
      return (LLR)((float)v * (c.csi[s] * c.inv));
    }
    int s;
    if (c.mod == 1) { // QPSK: LLRs 0,1 of a group of 4 take the second symbol's gain
      s = 2 * (b >> 2) + (((b & 3) < 2) ? 1 : 0);
    } else if (c.mod == 3) { // 64QAM: 0-3 first, 4,5 second, 6,7 first, 8-11 second
      const int r = b % 12;
      s = 2 * (b / 12) + ((r < 4 || r == 6 || r == 7) ? 0 : 1);
    } else {
      s = b / c.Qm;
    }
    const float f = rintf(c.csi[s] * c.scale);
    const int   w = f > 32767.0f ? 32767 : (int)f;
    return (LLR)(((int)v * w) >> 16);
  }
}


// grid = (ceil(w_stride/512), nof_sf*C): gather form of w[deint[i]] += e[i] (wrapping int16, rm_turbo.c:407-409).
// inv[j] = circular-buffer position n that lands on soft-buffer slot j (0xffffffff for padding); a thread owns two
// adjacent slots, sums their <= ceil(n_e/out_len) wraps from e and writes one dword: stores are coalesced and every
// slot, padding included, is written exactly once (no memset, no atomics).
// LLR = int8_t (srslte_rm_turbo_rx_lut_8bit, rm_turbo.c:428-465: wrapping int8 sums): a thread owns four adjacent slots.
template <typename LLR>
__global__ __launch_bounds__(256) void rm_rx_kernel(const LLR* __restrict__ e, LLR* __restrict__ w, const uint32_t* __restrict__ inv, RmGeom g)
{
  constexpr int PER = 4 / (int)sizeof(LLR); // slots per dword
  // the block: memory slot cbg = sf * g.C + cb; in grants mode its own (C, K, Qm, nof_re, table) come from the descriptor
  int             cbg = blockIdx.y, sf = cbg / g.C, cb = cbg - sf * g.C, C = g.C, Qm = g.Qm, out_len = g.out_len, w_len = g.w_stride, combine = g.combine;
  int             nre = g.nof_re[sf_class((g.tti0 + sf) % 10)], Nl = g.Nl, e_off = g.e_off;
  const uint32_t* tbl = inv;
  if (g.cbd) {
    const CbDesc d = g.cbd[blockIdx.y];
    sf = d.sf; cb = d.cb; cbg = sf * g.C + cb; C = d.C; Qm = d.Qm; out_len = 3 * d.K + 12; w_len = d.w_len; combine = d.combine; nre = d.nof_re; tbl = d.tbl;
    Nl = d.Nl; e_off = d.e_off;
  }
  const int j = PER * (blockIdx.x * blockDim.x + threadIdx.x);
  if (j >= w_len || (g.skip && combine && g.skip[cbg])) return;
  if (g.cb_ok_rst && !combine && j == 0) g.cb_ok_rst[cbg] = 0;
  const int QmL = Qm * Nl, Gp = nre / Nl; // Gp = nof_bits / (Qm N_L)
  const int gamma = Gp % C, n_e = QmL * (Gp / C);
  int       rp = cb * n_e, n_e2 = n_e;
  if (cb > C - gamma) { // sch.c:331-334 (the '>' quirk is upstream's)
    n_e2 = n_e + QmL;
    rp   = (C - gamma) * n_e + (cb - (C - gamma)) * n_e2;
  }
  const LLR* src = e + (size_t)sf * g.max_bits + e_off + rp;
  CsiW       cw;
  if (g.csi) cw = csi_setup(g, sf, nre, Qm);
  uint32_t   n[PER], word = 0;
  if constexpr (PER == 2) {
    const uint2 t = *reinterpret_cast<const uint2*>(tbl + j);
    n[0] = t.x; n[1] = t.y;
  } else {
    const uint4 t = *reinterpret_cast<const uint4*>(tbl + j);
    n[0] = t.x; n[1] = t.y; n[2] = t.z; n[3] = t.w;
  }
#pragma unroll
  for (int s = 0; s < PER; s++) {
    int acc = 0;
    if (n[s] != 0xffffffffu) {
      for (int i = (int)n[s]; i < n_e2; i += out_len) acc += g.csi ? csi_apply<LLR>(cw, src[i], rp + i) : src[i];
    }
    word |= ((uint32_t)acc & ((1u << (8 * sizeof(LLR))) - 1u)) << (8 * sizeof(LLR) * s);
  }
  uint32_t* dst = reinterpret_cast<uint32_t*>(w + (size_t)cbg * g.w_stride + j);
  *dst          = combine ? add_wrap<LLR>(*dst, word) : word;
}

// Same result with the code block's LLR segment staged in LDS: one workgroup per code block copies its n_e LLRs with 16-byte
// loads, then every thread produces 16 bytes of adjacent soft-buffer slots per step from LDS gathers. A 2-byte global gather
// costs the L1 one cache line per lane; the LDS gather a few bank-conflict cycles. The slot table is the 16-bit copy behind the
// 32-bit one (inv + w_stride), four steps' worth of it loaded ahead; LDS is sized to the segment (rm_lds_bytes) so that eight
// workgroups share a CU. Used when the segment fits 64 KB.
template <typename LLR>
__global__ __launch_bounds__(256) void rm_rx_lds_kernel(const LLR* __restrict__ e, LLR* __restrict__ w, const uint32_t* __restrict__ inv, RmGeom g)
{
  extern __shared__ __attribute__((aligned(16))) char seg_raw[];
  constexpr int PER = 16 / (int)sizeof(LLR), NV = PER / 8; // slots per 16 bytes; uint4 loads of 16-bit table entries per step
  LLR*          seg = reinterpret_cast<LLR*>(seg_raw);
  int             cbg = blockIdx.x, sf = cbg / g.C, cb = cbg - sf * g.C, C = g.C, Qm = g.Qm, out_len = g.out_len, w_len = g.w_stride, combine = g.combine;
  int             nre = g.nof_re[sf_class((g.tti0 + sf) % 10)], Nl = g.Nl, e_off = g.e_off;
  const uint32_t* tbl = inv;
  if (g.cbd) {
    const CbDesc d = g.cbd[blockIdx.x];
    sf = d.sf; cb = d.cb; cbg = sf * g.C + cb; C = d.C; Qm = d.Qm; out_len = 3 * d.K + 12; w_len = d.w_len; combine = d.combine; nre = d.nof_re; tbl = d.tbl;
    Nl = d.Nl; e_off = d.e_off;
  }
  if (g.skip && combine && g.skip[cbg]) return;
  if (g.cb_ok_rst && !combine && threadIdx.x == 0) g.cb_ok_rst[cbg] = 0;
  const int QmL = Qm * Nl, Gp = nre / Nl;
  const int gamma = Gp % C, n_e = QmL * (Gp / C);
  int       rp = cb * n_e, n_e2 = n_e;
  if (cb > C - gamma) { // sch.c:331-334
    n_e2 = n_e + QmL;
    rp   = (C - gamma) * n_e + (cb - (C - gamma)) * n_e2;
  }
  const LLR* src = e + (size_t)sf * g.max_bits + e_off + rp;
  // the segment starts at an arbitrary LLR index: copy from the 16-byte boundary below it
  const int mis = (int)((reinterpret_cast<uintptr_t>(src) & 15) / sizeof(LLR));
  const int4* s4 = reinterpret_cast<const int4*>(src - mis);
  const int   n16 = (n_e2 + mis + PER - 1) / PER;
  if (g.csi) { // weigh while staging: element j of 16-byte word i is LLR rp - mis + PER * i + j of the subframe
    const CsiW cw = csi_setup(g, sf, nre, Qm);
    for (int i = threadIdx.x; i < n16; i += 256) {
      union {
        int4 v;
        LLR  h[PER];
      } u;
      u.v = s4[i];
#pragma unroll
      for (int j = 0; j < PER; j++) u.h[j] = csi_apply<LLR>(cw, u.h[j], rp - mis + PER * i + j);
      reinterpret_cast<int4*>(seg)[i] = u.v;
    }
  } else {
#pragma unroll 4
    for (int i = threadIdx.x; i < n16; i += 256) reinterpret_cast<int4*>(seg)[i] = s4[i];
  }
  __syncthreads();
  const LLR*   ls    = seg + mis;
  const uint4* inv16 = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(tbl + w_len));
  uint4*       dst   = reinterpret_cast<uint4*>(w + (size_t)cbg * g.w_stride);
  const int    ngroups = w_len / PER;
  for (int j0 = threadIdx.x; j0 < ngroups; j0 += 4 * 256) {
    uint4 tt[4][NV];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = j0 + 256 * u;
#pragma unroll
      for (int v = 0; v < NV; v++) tt[u][v] = j < ngroups ? inv16[j * NV + v] : make_uint4(~0u, ~0u, ~0u, ~0u);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = j0 + 256 * u;
      if (j >= ngroups) break;
      uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < PER; s++) {
        const uint32_t pair = (&tt[u][s / 8].x)[(s / 2) & 3], n = (s & 1) ? pair >> 16 : pair & 0xffffu;
        int            acc  = 0;
        if (n != 0xffffu) {
          for (int i = (int)n; i < n_e2; i += out_len) acc += ls[i];
        }
        constexpr int BITS = 8 * (int)sizeof(LLR);
        o[s * BITS / 32] |= ((uint32_t)acc & ((1u << BITS) - 1u)) << ((s * BITS) & 31);
      }
      if (combine) {
        const uint4 old = dst[j];
        dst[j] = make_uint4(add_wrap<LLR>(old.x, o[0]), add_wrap<LLR>(old.y, o[1]), add_wrap<LLR>(old.z, o[2]), add_wrap<LLR>(old.w, o[3]));
      } else {
        dst[j] = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

__device__ __forceinline__ uint32_t gf24_mul(uint32_t a, uint32_t b, uint32_t poly)
{ // a(x) b(x) mod g(x), deg g = 24 (poly carries the x^24 term)
  uint32_t r = 0;
  for (int i = 23; i >= 0; i--) {
    r <<= 1;
    if (r & 0x1000000u) r ^= poly;
    if ((b >> i) & 1) r ^= a;
  }
  return r;
}

// CRC24 (init 0, MSB first) of nbytes bytes by a 256-thread block: the message is zero-extended at the FRONT to 256 equal chunks
// (leading zeros do not change the remainder), every thread runs the byte-table recursion over its chunk, and the chunk
// remainders are folded pairwise with x^(8*chunk*2^level) mod g. tab/red: 256 words of LDS each. Result valid in every thread.
template <typename Byte>
__device__ uint32_t block_crc24(Byte byte_at, int nbytes, uint32_t poly, uint32_t* tab, uint32_t* red)
{
  const int t = threadIdx.x;
  {
    uint32_t v = (uint32_t)t << 16;
    for (int i = 0; i < 8; i++) {
      v <<= 1;
      if (v & 0x1000000u) v ^= poly;
    }
    tab[t] = v;
  }
  __syncthreads();
  const int cB = (nbytes + 255) / 256, pad = 256 * cB - nbytes;
  uint32_t  crc = 0;
  for (int i = 0; i < cB; i++) {
    const int v = t * cB + i - pad;
    if (v >= 0) crc = ((crc << 8) & 0xffffffu) ^ tab[((crc >> 16) & 0xff) ^ byte_at(v)];
  }
  uint32_t m = 1; // x^(8 cB) mod g
  for (int i = 0; i < 8 * cB; i++) {
    m <<= 1;
    if (m & 0x1000000u) m ^= poly;
  }
  red[t] = crc;
  __syncthreads();
  for (int s = 1; s < 256; s <<= 1) {
    uint32_t v = 0;
    if ((t & (2 * s - 1)) == 0) v = gf24_mul(red[t], m, poly) ^ red[t + s];
    __syncthreads();
    if ((t & (2 * s - 1)) == 0) red[t] = v;
    m = gf24_mul(m, m, poly);
    __syncthreads();
  }
  return red[0];
}

struct TbGeom {
  const SfDesc* desc; // grants mode (tb_crc_bytes_kernel): per-subframe (C, K, tbs, rlen), C = code-block slots per subframe; else null
  int C, K, tbs, rlen, cb_stride, tb_stride;
  int nof_sf, cw1_off; // grants mode: subframes of the call, descriptor offset of their second transport blocks
};

// one workgroup per subframe: assemble the payload (sch.c:360,:401-410) and check CRC24A (sch.c:470-488).
// The CRC is the XOR over set bits of precomputed x^(n-1-j) mod g: branch-free, every lookup independent.
__global__ __launch_bounds__(512) void tb_crc_kernel(const uint8_t* __restrict__ cb_bytes, const uint8_t* __restrict__ cb_ok,
                                                     const uint32_t* __restrict__ crc_rem, uint8_t* __restrict__ tb, uint8_t* __restrict__ tb_ok,
                                                     TbGeom g)
{
  __shared__ uint32_t red[8];
  const int C = g.C, K = g.K, tbs = g.tbs, rlen = g.rlen;
  const int sf = blockIdx.x, nbytes = tbs / 8 + 3, rb = rlen / 8;
  uint8_t*  dst = tb + (size_t)sf * g.tb_stride;
  uint32_t  syn = 0;
  for (int b0 = threadIdx.x * 4; b0 < nbytes + 3; b0 += blockDim.x * 4) {
    uint32_t word = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int b = b0 + t;
      int       cb = b / rb;
      if (cb > C - 1) cb = C - 1;
      const int     off = b - cb * rb;
      const uint8_t v   = (b < nbytes + 3 && off < K / 8) ? cb_bytes[((size_t)sf * g.C + cb) * g.cb_stride + off] : 0;
      if (b < nbytes + 3) dst[b] = v;
      word |= (uint32_t)(b < nbytes ? v : 0) << (8 * t);
    }
#pragma unroll
    for (int j = 0; j < 32; j++) { // bit j of byte t = message bit 8*(b0+t) + (7 - j%8)
      const int      t = j >> 3, bit = 8 * (b0 + t) + 7 - (j & 7);
      const uint32_t m = 0u - ((word >> j) & 1u);
      syn ^= (bit < 8 * nbytes ? crc_rem[bit] : 0u) & m;
    }
  }
  for (int o = 32; o > 0; o >>= 1) syn ^= __shfl_xor(syn, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = syn;
  __syncthreads();
  if (threadIdx.x == 0) {
    syn = 0;
    for (int i = 0; i < (int)blockDim.x / 64; i++) syn ^= red[i];
    bool ok = syn == 0;
    for (int c = 0; c < C; c++) ok = ok && cb_ok[sf * g.C + c];
    // par_rx == par_tx && par_rx != 0 (sch.c:481): a zero parity with zero syndrome is rejected upstream
    const uint8_t* p = dst + tbs / 8;
    ok               = ok && (p[0] | p[1] | p[2]);
    tb_ok[sf]        = ok ? 1 : 0;
  }
}

// The same for transport blocks of any size (grants mode): the bytes go through LDS, 256 threads take the CRC24A of contiguous chunks with a
// byte table, and the chunk CRCs are combined with 256 weights x^(8 n) mod g that the host makes once per transport block size.
constexpr int TB_MAX_BITS = 105528;
__global__ __launch_bounds__(256) void tb_crc_bytes_kernel(const uint8_t* __restrict__ cb_bytes, const uint8_t* __restrict__ cb_ok, uint8_t* __restrict__ tb,
                                                           uint8_t* __restrict__ tb_ok, TbGeom g)
{
  __shared__ uint32_t tab[256], red[4];
  __shared__ uint8_t  bytes[TB_MAX_BITS / 8 + 8]; // the largest one-layer transport block (36.213 Table 7.1.7.2.1-1: 105528 bits at 110 PRB) + CRC
  // row r < nof_sf: transport block 0 of subframe r (descriptor / HARQ slot r); row nof_sf + b: transport block 1 of subframe b (slot cw1_off + b)
  const int    row = blockIdx.x, sf = row < g.nof_sf ? row : g.cw1_off + row - g.nof_sf;
  const SfDesc d   = g.desc[sf];
  const int    C = d.C, K = d.K, nbytes = d.tbs / 8 + 3, rb = d.rlen / 8, t = threadIdx.x;
  uint8_t*     dst = tb + (size_t)row * g.tb_stride;
  if (C == 0) { // no transport block here
    if (t == 0) tb_ok[row] = 0;
    return;
  }
  {
    uint32_t v = (uint32_t)t << 16; // byte table of CRC24A
    for (int i = 0; i < 8; i++) {
      v <<= 1;
      if (v & 0x1000000u) v ^= 0x1864CFBu;
    }
    tab[t] = v;
  }
  for (int cb = 0; cb < C; cb++) { // block cb carries bytes [cb rb, (cb + 1) rb) of the transport block (the last one also the 3 bytes behind)
    const uint8_t* src = cb_bytes + ((size_t)sf * g.C + cb) * g.cb_stride;
    const int      n   = cb == C - 1 ? nbytes + 3 - cb * rb : rb;
    for (int i = t; i < n; i += 256) {
      const uint8_t v = i < K / 8 ? src[i] : 0;
      dst[cb * rb + i] = v;
      if (cb * rb + i < nbytes) bytes[cb * rb + i] = v;
    }
  }
  __syncthreads();
  // CRC of 256 contiguous chunks of cB bytes (zeros in front do not change a CRC), each weighted by x^(8 cB (chunks behind it)) mod g
  const int cB = (nbytes + 255) / 256, pad = 256 * cB - nbytes;
  uint32_t  crc = 0;
  for (int i = 0; i < cB; i++) {
    const int v = t * cB + i - pad;
    if (v >= 0) crc = ((crc << 8) & 0xffffffu) ^ tab[((crc >> 16) & 0xff) ^ bytes[v]];
  }
  uint32_t syn = gf24_mul(crc, d.crc_fac[t], 0x1864CFBu);
  for (int o = 32; o > 0; o >>= 1) syn ^= __shfl_xor(syn, o, 64);
  if ((t & 63) == 0) red[t >> 6] = syn;
  __syncthreads();
  if (t == 0) {
    bool ok = (red[0] ^ red[1] ^ red[2] ^ red[3]) == 0;
    for (int c = 0; c < C; c++) ok = ok && cb_ok[sf * g.C + c];
    ok        = ok && (bytes[nbytes - 3] | bytes[nbytes - 2] | bytes[nbytes - 1]); // par_rx != 0 (sch.c:481)
    tb_ok[row] = ok ? 1 : 0;
  }
}

// ---- grants mode: RE lists and scrambling sequences made on the device from the grants of the batch
// pdsch.c:81-206 as a per-RE rule for a single-port cell (see pdsch_re_indices below and oracle/orc_pdsch.c): symbol sym = 7 s + l,
// sub-carrier k. q_off: what upstream's `offset` variable holds when it reaches the half PRBs of an odd-bandwidth cell (pdsch.c:172-190)
__host__ __device__ __forceinline__ bool pdsch_re_used(int P, int cell_id, int sf_idx, int q_off, int s, int l, int k, int nof_ports = 1)
{
  const int  nre  = 12 * P;
  const bool sync = (s == 0 && (sf_idx == 0 || sf_idx == 5) && l >= 5) || (s == 1 && sf_idx == 0 && l < 4);
  if (sync && k + 36 >= nre / 2 && k < nre / 2 + 36) return false;
  if (l == 0 || l == 4 || (l == 1 && nof_ports == 4)) { // phy_common.h:139-141
    const int  p      = k / 12;
    const bool centre = p >= P / 2 - 3 && p < P / 2 + 3 + (P % 2);
    if (nof_ports == 1) {
      const int off = (centre && sync) ? q_off : (l == 0 ? cell_id % 6 : (cell_id + 3) % 6);
      if (k % 6 == off % 6) return false;
    } else { // every port's CRS positions stay empty: one RE in three, the same offset in every CRS symbol (pdsch.c:103-107)
      const int off = (centre && sync) ? (l == 1 ? (q_off >> 8) : (q_off & 255)) : cell_id % 3; // bits 8..: the value by symbol 1 of slot 1 (4 ports)
      if (k % 3 == off % 3) return false;
    }
  }
  return true;
}

// One workgroup per subframe: idx_out[sf][...] = the grid positions of the subframe's PDSCH REs in the reference's order (symbol-major,
// sub-carrier ascending). Two sweeps over the 14 x 12 P grid positions in chunks of 64: per-chunk counts by ballot, a prefix sum, then
// every wavefront writes its chunks' positions compacted (coalesced).
constexpr int RELIST_THREADS = 512;
__global__ __launch_bounds__(RELIST_THREADS) void pdsch_relist_kernel(const GrantDev* __restrict__ gr, uint32_t* __restrict__ idx_out, int P, int cell_id,
                                                                      int max_re, int nof_ports)
{
  __shared__ int cnt[14 * 21 + 1]; // chunks: 14 symbols x ceil(12 * 110 / 64)
  const int      sf = blockIdx.x, nre = 12 * P, cps = (nre + 63) / 64, nchunks = 14 * cps;
  const int      wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = RELIST_THREADS / 64;
  const GrantDev g  = gr[sf];
  auto           used = [&](int sym, int kc) { // position (sym, kc * 64 + lane)
    const int k = kc * 64 + lane, s = sym >= 7 ? 1 : 0, l = sym - 7 * s, p = k / 12;
    if (k >= nre || !((g.mask[s][p >> 5] >> (p & 31)) & 1u)) return false;
    return pdsch_re_used(P, cell_id, g.sf_idx, g.q_off, s, l, k, nof_ports);
  };
  __shared__ unsigned long long msk[14 * 21]; // the chunks' masks: the second sweep does not evaluate the rule again
  for (int sym = 0; sym < 14; sym++) {
    const bool on = sym >= 7 || sym >= g.lstart; // control region
    for (int kc = wave; kc < cps; kc += nwaves) {
      const unsigned long long b = on ? __ballot(used(sym, kc)) : 0ull;
      if (lane == 0) {
        cnt[sym * cps + kc] = __popcll(b);
        msk[sym * cps + kc] = b;
      }
    }
  }
  __syncthreads();
  if (wave == 0) { // exclusive prefix over <= 294 entries: five consecutive entries per lane, then a wavefront scan
    constexpr int PL = (14 * 21 + 63) / 64;
    int           v[PL], sum = 0;
#pragma unroll
    for (int j = 0; j < PL; j++) {
      const int c = lane * PL + j;
      v[j]        = c < nchunks ? cnt[c] : 0;
      sum += v[j];
    }
    int incl = sum;
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    int acc = incl - sum;
#pragma unroll
    for (int j = 0; j < PL; j++) {
      const int c = lane * PL + j;
      if (c < nchunks) cnt[c] = acc;
      acc += v[j];
    }
  }
  __syncthreads();
  uint32_t* o = idx_out + (size_t)sf * max_re;
  for (int sym = (g.lstart < 7 ? g.lstart : 7); sym < 14; sym++) {
    for (int kc = wave; kc < cps; kc += nwaves) {
      const unsigned long long b = msk[sym * cps + kc];
      if ((b >> lane) & 1ull) o[cnt[sym * cps + kc] + __popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)(sym * nre + kc * 64 + lane);
    }
  }
}
__global__ __launch_bounds__(256) void scr_gen_kernel(const GrantDev* __restrict__ gr, const uint32_t* __restrict__ basis, uint32_t* __restrict__ out,
                                                      int words, int cell_id)
{
  const int sf = blockIdx.y, w = blockIdx.x * 256 + threadIdx.x;
  if (w >= words) return;
  const GrantDev g = gr[sf];
  uint32_t       c_init = ((uint32_t)g.rnti << 14) + ((uint32_t)g.cw << 13) + ((uint32_t)g.sf_idx << 9) + (uint32_t)cell_id, v = basis[w];
  for (int j = 0; j < 31; j++) {
    if ((c_init >> j) & 1u) v ^= basis[(size_t)(1 + j) * words + w];
  }
  out[(size_t)sf * words + w] = v;
}

// Same result from the per-block syndrome shares the windowed turbo decoders emit (tdec_set_tb_syndrome): the CRC is linear,
// so the TB syndrome is the XOR of the C shares; what is left is the payload copy.
__global__ __launch_bounds__(256) void tb_asm_kernel(const uint8_t* __restrict__ cb_bytes, const uint8_t* __restrict__ cb_ok,
                                                     const uint32_t* __restrict__ cb_syn, uint8_t* __restrict__ tb, uint8_t* __restrict__ tb_ok,
                                                     TbGeom g)
{
  const int sf = blockIdx.x, nbytes = g.tbs / 8 + 3, rb = g.rlen / 8;
  uint8_t*  dst = tb + (size_t)sf * g.tb_stride;
  auto      src = [&](int b) -> uint8_t {
    int cb = b / rb;
    if (cb > g.C - 1) cb = g.C - 1;
    const int off = b - cb * rb;
    return off < g.K / 8 ? cb_bytes[((size_t)sf * g.C + cb) * g.cb_stride + off] : (uint8_t)0;
  };
  for (int b = threadIdx.x; b < nbytes + 3; b += blockDim.x) dst[b] = src(b);
  if (threadIdx.x == 0) {
    uint32_t syn = 0;
    bool     ok  = true;
    for (int c = 0; c < g.C; c++) {
      syn ^= cb_syn[sf * g.C + c];
      ok = ok && cb_ok[sf * g.C + c];
    }
    // par_rx == par_tx && par_rx != 0 (sch.c:481): a zero parity with zero syndrome is rejected upstream
    ok        = ok && syn == 0 && (src(g.tbs / 8) | src(g.tbs / 8 + 1) | src(g.tbs / 8 + 2));
    tb_ok[sf] = ok ? 1 : 0;
  }
}

// largest segment of any subframe class + the Qm extra LLRs of the last blocks must fit the LDS kernel; w_stride is a multiple of 32
int rm_lds_bytes(const RmGeom& g, int llr_bytes)
{
  int mx = g.nof_re[0] > g.nof_re[1] ? g.nof_re[0] : g.nof_re[1];
  mx     = mx > g.nof_re[2] ? mx : g.nof_re[2];
  return ((g.Qm * (mx / g.C) + 2 * g.Qm) * llr_bytes + 32 + 15) & ~15; // + the bytes below the 16-byte boundary and the rounded-up last load
}
bool rm_fits_lds(const RmGeom& g, int llr_bytes = 2) { return rm_lds_bytes(g, llr_bytes) <= 64 * 1024; }

// slot -> circular-buffer position: 32-bit entries [w_stride] for the generic kernels, then the same as 16-bit entries for the LDS kernel
std::vector<uint32_t> rm_slot_table(const std::vector<uint32_t>& t, uint32_t w_stride)
{
  std::vector<uint32_t> inv(w_stride + w_stride / 2, 0xffffffffu);
  uint16_t*             inv16 = reinterpret_cast<uint16_t*>(inv.data() + w_stride);
  for (uint32_t n = 0; n < t.size(); n++) {
    inv[t[n]]   = n;
    inv16[t[n]] = (uint16_t)n; // n < 3 * 6144 + 12
  }
  return inv;
}

// pdsch.c:81-206 as a per-RE rule (see oracle/orc_pdsch.c for the derivation): symbol-major, sub-carrier ascending,
// skipping CRS, and the central 72 sub-carriers of the PSS/SSS symbols (slot 0, l >= 5, sf 0/5) and PBCH symbols (slot 1, l < 4, sf 0)
void pdsch_re_indices(uint32_t cell_id, uint32_t nof_prb, uint32_t nof_ports, uint32_t sf_idx, uint32_t lstart, std::vector<uint32_t>& idx)
{
  const uint32_t nre = 12 * nof_prb, step = nof_ports == 1 ? 6 : 3; // 2/4 ports: the other ports' CRS positions are left empty too (pdsch.c:103-107)
  idx.clear();
  for (uint32_t s = 0; s < 2; s++) {
    for (uint32_t l = (s == 0 ? lstart : 0); l < 7; l++) {
      const bool     has_ref = l == 0 || l == 4 || (l == 1 && nof_ports == 4); // phy_common.h:139-141
      const uint32_t offset  = l == 0 ? cell_id % 6 : (cell_id + 3) % 6;
      const bool     sync    = (s == 0 && (sf_idx == 0 || sf_idx == 5) && l >= 5) || (s == 1 && sf_idx == 0 && l < 4);
      for (uint32_t k = 0; k < nre; k++) {
        if (sync && k + 36 >= nre / 2 && k < nre / 2 + 36) continue;
        if (has_ref && (k % step) == offset % step) continue;
        idx.push_back((s * 7 + l) * nre + k);
      }
    }
  }
}

template <typename T>
int upload(T** d, const std::vector<T>& h)
{
  HIP_TRY(hipMalloc((void**)d, sizeof(T) * (h.size() ? h.size() : 1)));
  HIP_TRY(hipMemcpy(*d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
  return SRSLTE_SUCCESS;
}

} // namespace

// Device / host resources of the per-subframe-grant mode (srslte_hip_dl_rx_batch_grants)
struct GrantsState {
  srslte_hip_tdec_t* tdec;       // any block length up to 6144
  uint32_t           Cmax, stride, max_re, max_bits, words;
  uint32_t           V; // per-subframe slots: max_batch, twice that on a cell where two-layer grants can occur (codeword 1 of subframe b: slot max_batch + b)
  uint32_t *         d_relist, *d_scr, *d_basis, *d_cb_iters;
  int16_t *          d_e, *d_w;
  uint8_t *          d_cb_bytes, *d_cb_ok, *d_desc;
  float*             d_csi;     // [B][max_re], cfg.csi_enable
  uint32_t*          d_csi_max; // [B]
  size_t             desc_bytes;
  // descriptors of a call are built in one of four pinned host buffers and copied asynchronously: the host does not wait for the stream
  uint8_t*   h_pin[4];
  hipEvent_t h_ev[4];
  bool       h_used[4];
  uint32_t   h_slot;
  std::map<std::pair<uint32_t, uint32_t>, uint32_t*>   rm_tbl; // (K, rv) -> slot table in the layout of that K's decoder
  std::map<uint32_t, uint32_t*>                        crc_fac; // tbs -> tb_crc_bytes_kernel's 256 chunk weights
};

struct srslte_hip_dl_rx {
  srslte_hip_dl_rx_cfg_t cfg;
  srslte_hip_ofdm_t*     ofdm;
  srslte_hip_chest_dl_t* chest;
  srslte_hip_tdec_t*     tdec;
  srslte_hip_cbsegm_t    seg;
  PdschGeom              pg;
  RmGeom                 rg;
  TbGeom                 tg;
  uint32_t               W, in_stride;
  uint32_t*              d_idx[3];
  uint32_t*              d_scr;
  uint32_t*              d_rm_tbl;      // rv 0
  uint32_t*              d_rm_tbl_rv[4]; // [0] aliases d_rm_tbl; 1..3 built on first use (srslte_hip_dl_rx_batch_harq)
  uint32_t               harq_rv;
  int                    harq_combine;
  uint32_t*              d_tbcrc;
  cf32 *                 d_grid, *d_ce, *d_d;
  ChestResDev*           d_res;
  int16_t *              d_e, *d_w;
  uint8_t *              d_cb_bytes, *d_cb_ok;
  uint32_t*              d_cb_iters;
  uint32_t *             d_tb_rem, *d_cb_syn; // TB CRC shares from the windowed decoders ([C][K] table, [B*C] out); null for W = 0
  float*                 d_csi;     // [B][max_re], cfg.csi_enable
  uint32_t*              d_csi_max; // [B]
  const cf32*            grid_in; // resource grids supplied by the caller (srslte_hip_dl_rx_grid_batch) instead of d_grid
  struct GrantsState*    gs;      // srslte_hip_dl_rx_batch_grants: created on first use
  struct srslte_hip_dl_rx* cw1;   // two-layer modes: the second codeword's back end (rate de-matching, decoder, TB assembly and their buffers)
};

static void grants_free(GrantsState* g);

extern "C" void srslte_hip_dl_rx_destroy(srslte_hip_dl_rx_t* q)
{
  if (!q) return;
  srslte_hip_dl_rx_destroy(q->cw1);
  srslte_hip_ofdm_destroy(q->ofdm);
  srslte_hip_chest_dl_destroy(q->chest);
  srslte_hip_tdec_destroy(q->tdec);
  void* bufs[] = {q->d_idx[0], q->d_idx[1], q->d_idx[2], q->d_scr, q->d_rm_tbl, q->d_tbcrc, q->d_grid, q->d_ce, q->d_d,
                  q->d_res,    q->d_e,      q->d_w,      q->d_cb_bytes, q->d_cb_ok, q->d_cb_iters, q->d_tb_rem, q->d_cb_syn,
                  q->d_csi,    q->d_csi_max, q->d_rm_tbl_rv[1], q->d_rm_tbl_rv[2], q->d_rm_tbl_rv[3]};
  for (void* b : bufs) {
    if (b) (void)hipFree(b);
  }
  grants_free(q->gs);
  delete q;
}

// rate de-matching table of redundancy version rv in the decoder's input layout (rm_turbo.c:160-260)
static int dl_rx_rm_table(srslte_hip_dl_rx_t* q, uint32_t rv, uint32_t** d_tbl)
{
  const uint32_t        K = q->seg.K1;
  std::vector<uint32_t> t;
  lte_rm_rx_table(K, rv, t);
  if (q->W) {
    for (auto& v : t) {
      v = v < 3 * K ? (v % 3) * (K + 32) + ((v / 3) % (K / q->W)) * q->W + (v / 3) / (K / q->W) : (v - 3 * K) + 3 * (K + 32);
    }
  }
  return upload(d_tbl, rm_slot_table(t, q->in_stride)); // in_stride is a multiple of 32
}

// cw: 0 = a whole pipeline; 1 = the back end of the second codeword of a two-layer mode (cfg->mod / tbs already those of that codeword):
// no OFDM / estimator objects, no grids, scrambling sequence q = 1 (36.211 6.3.1)
static srslte_hip_dl_rx_t* dl_rx_create_impl(const srslte_hip_dl_rx_cfg_t* cfg, int cw)
{
  if (!cfg || cfg->max_batch == 0 || cfg->mod < 1 || cfg->mod > 4 || cfg->max_iterations == 0 || cfg->nof_rx_antennas > 4 || cfg->nof_ports > 4 ||
      cfg->nof_ports == 3 || (cfg->nof_ports == 4 && cfg->chest_cfg.interpolate_subframe)) {
    hip_log("[srslte_hip] dl_rx: invalid configuration\n");
    return nullptr;
  }
  const bool mimo = cfg->tx_scheme != 0;
  if (mimo && cw == 0) { // what ra_dl.c:556-600 and precoding.c:1710-1759,:1087-1114 let through
    const bool two = cfg->tbs2 != 0;
    const bool ok  = (cfg->tx_scheme == 2 || cfg->tx_scheme == 3) && cfg->nof_ports == 2 && cfg->nof_rx_antennas == 2 && !cfg->llr_8bit &&
                    (cfg->tx_scheme == 2 || two) && (two ? cfg->pmi < 2 : cfg->pmi < 4) && (!two || (cfg->mod2 >= 1 && cfg->mod2 <= 4));
    if (!ok) {
      hip_log("[srslte_hip] dl_rx: two-layer modes need a 2-port cell, 2 receive antennas, 16-bit LLRs; CDD with two transport blocks, "
              "multiplexing with two (pmi 0-1) or one (pmi 0-3)\n");
      return nullptr;
    }
  }
  auto* q = new srslte_hip_dl_rx();
  memset(q, 0, sizeof(*q));
  q->cfg = *cfg;
  if (srslte_hip_cbsegm(&q->seg, cfg->tbs) || q->seg.F || q->seg.C2 || (cfg->tbs % 8)) {
    hip_log("[srslte_hip] dl_rx: TBS %u needs filler bits or two code-block sizes; not supported on device yet\n", cfg->tbs);
    delete q;
    return nullptr;
  }
  const uint32_t P = cfg->nof_prb, nre = 12 * P, B = cfg->max_batch, C = q->seg.C, K = q->seg.K1, Qm = 2 * (uint32_t)cfg->mod;
  const uint32_t lstart = cfg->cfi + (P < 10 ? 1 : 0); // SRSLTE_NOF_CTRL_SYMBOLS, phy_common.h:143
  const uint32_t nrx    = cfg->nof_rx_antennas ? cfg->nof_rx_antennas : 1;
  const uint32_t npt    = cfg->nof_ports ? cfg->nof_ports : 1;
  if (cw == 0) {
    q->ofdm  = srslte_hip_ofdm_create((int)P, 1, 1);
    q->chest = srslte_hip_chest_dl_create(cfg->cell_id, P, npt, 1);
  }
  q->tdec  = srslte_hip_tdec_create(K, B * C);
  bool ok  = (cw || (q->ofdm && q->chest)) && q->tdec;
  // RE lists
  uint32_t max_re = 0;
  const uint32_t rep_sf[3] = {0, 5, 1};
  for (int c = 0; c < 3 && ok; c++) {
    std::vector<uint32_t> idx;
    pdsch_re_indices(cfg->cell_id, P, npt, rep_sf[c], lstart, idx);
    q->pg.cls[c].nof_re = (int)idx.size();
    q->rg.nof_re[c]     = (int)idx.size();
    max_re              = idx.size() > max_re ? (uint32_t)idx.size() : max_re;
    if (cw) continue; // the lists are the first codeword's object's
    ok                  = upload(&q->d_idx[c], idx) == SRSLTE_SUCCESS;
    q->pg.cls[c].idx    = q->d_idx[c];
  }
  const uint32_t max_bits = (max_re * Qm + 15) & ~15u, scr_words = (max_re * Qm + 31) / 32 + 1; // spare word: the demapper reads two per RE
  // scrambling sequences, one per subframe index (sequences.c:58-60, pdsch.c:469)
  if (ok) {
    std::vector<uint32_t> scr((size_t)10 * scr_words, 0);
    std::vector<uint8_t>  c;
    for (uint32_t sf = 0; sf < 10; sf++) {
      lte_gold_sequence(((uint32_t)cfg->rnti << 14) + ((uint32_t)cw << 13) + (sf << 9) + cfg->cell_id, max_re * Qm, c);
      for (uint32_t i = 0; i < max_re * Qm; i++) scr[(size_t)sf * scr_words + (i >> 5)] |= (uint32_t)c[i] << (i & 31);
    }
    ok = upload(&q->d_scr, scr) == SRSLTE_SUCCESS;
  }
  // rate-dematching table in the decoder's input layout (rm_turbo.c:160-260)
  q->W         = cfg->llr_8bit ? srslte_hip_tdec_autoimp_get_subblocks_8bit(K) : srslte_hip_tdec_autoimp_get_subblocks(K);
  q->in_stride = (srslte_hip_tdec_input_len(K, q->W != 0) + 31) & ~31u;
  if (ok) {
    ok                = dl_rx_rm_table(q, 0, &q->d_rm_tbl) == SRSLTE_SUCCESS;
    q->d_rm_tbl_rv[0] = q->d_rm_tbl;
  }
  // TB CRC24A remainders x^(tbs+24-1-j) mod g
  if (ok) {
    std::vector<uint32_t> rem(cfg->tbs + 24);
    uint32_t              v = 1;
    for (int j = (int)cfg->tbs + 23; j >= 0; j--) {
      rem[j] = v;
      v <<= 1;
      if (v & 0x1000000) v ^= 0x1864CFB;
    }
    ok = upload(&q->d_tbcrc, rem) == SRSLTE_SUCCESS;
    if (ok && q->W) { // per code block, in the decoder's array order (window-interleaved); 0 on the CB CRC bits
      const uint32_t        rlen = C == 1 ? K : K - 24, Lw = K / q->W;
      std::vector<uint32_t> t((size_t)C * K, 0);
      for (uint32_t c = 0; c < C; c++) {
        for (uint32_t n = 0; n < rlen; n++) {
          const uint32_t pos = c * rlen + n;
          if (pos < cfg->tbs + 24) t[(size_t)c * K + (n % Lw) * q->W + n / Lw] = rem[pos];
        }
      }
      ok = upload(&q->d_tb_rem, t) == SRSLTE_SUCCESS && hipMalloc((void**)&q->d_cb_syn, sizeof(uint32_t) * B * C) == hipSuccess;
    }
  }
  const size_t glen = (size_t)14 * nre;
  if (cw == 0) {
    ok = ok && hipMalloc((void**)&q->d_grid, sizeof(cf32) * glen * B * nrx) == hipSuccess &&
         hipMalloc((void**)&q->d_ce, sizeof(cf32) * glen * B * nrx * npt) == hipSuccess &&
         hipMalloc((void**)&q->d_res, sizeof(ChestResDev) * B) == hipSuccess;
  }
  ok = ok && hipMalloc((void**)&q->d_e, sizeof(int16_t) * ((size_t)max_bits * B + 16)) == hipSuccess /* +16: rm_rx_lds_kernel reads whole 16-byte words */ &&
       hipMalloc((void**)&q->d_w, sizeof(int16_t) * (size_t)q->in_stride * B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_cb_bytes, (size_t)(K / 8) * B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_cb_ok, (size_t)B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_cb_iters, sizeof(uint32_t) * B * C) == hipSuccess;
  if (ok && cfg->csi_enable) {
    ok = hipMalloc((void**)&q->d_csi, sizeof(float) * (size_t)max_re * B) == hipSuccess &&
         hipMalloc((void**)&q->d_csi_max, sizeof(uint32_t) * B) == hipSuccess;
  }
  // HARQ state of slots that have not seen new data yet (a retransmission into such a slot combines with an empty soft buffer and
  // decodes every block); the memsets run on the null stream, which the callers' non-blocking streams do not order against: wait here
  ok = ok && hipMemset(q->d_cb_ok, 0, (size_t)B * C) == hipSuccess && hipMemset(q->d_w, 0, sizeof(int16_t) * (size_t)q->in_stride * B * C) == hipSuccess &&
       hipMemset(q->d_cb_bytes, 0, (size_t)(K / 8) * B * C) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  if (!ok) {
    hip_log("[srslte_hip] dl_rx: initialisation failed\n");
    srslte_hip_dl_rx_destroy(q);
    return nullptr;
  }
  q->pg.grid_len = (int)glen; q->pg.max_re = (int)max_re; q->pg.max_bits = (int)max_bits; q->pg.mod = cfg->mod; q->pg.Qm = (int)Qm;
  q->pg.mmse = cfg->mmse; q->pg.scr_words = (int)scr_words; q->pg.nof_rx = (int)nrx; q->pg.nof_ports = (int)npt;
  q->pg.csi = q->d_csi; q->pg.csi_max = q->d_csi_max;
  // apply_power_allocation (pdsch.c:518-554) with rho_b = 1: pdsch_scaling = rho_a = 10^(p_a/20), times sqrt(2) for a 2-port cell
  q->pg.inv_scaling = cfg->power_scale ? 1.0f / (powf(10.0f, cfg->p_a / 20.0f) * (npt == 1 ? 1.0f : sqrtf(2.0f))) : 1.0f;
  // code-block split in units of Qm N_L bits: N_L = 2 for transmit diversity, 1 where nof_layers == nof_tb (srslte_dlsch_decode2, sch.c:507-531)
  q->rg.csi = q->d_csi; q->rg.csi_max = q->d_csi_max; q->rg.max_re = (int)max_re; q->rg.mod = cfg->mod; q->rg.Nl = (npt > 1 && !mimo) ? 2 : 1;
  q->rg.C = (int)C; q->rg.K = (int)K; q->rg.Qm = (int)Qm; q->rg.max_bits = (int)max_bits; q->rg.w_stride = (int)q->in_stride;
  q->rg.out_len = (int)(3 * K + 12);
  q->tg.C = (int)C; q->tg.K = (int)K; q->tg.tbs = (int)cfg->tbs; q->tg.rlen = (int)(C == 1 ? K : K - 24); q->tg.cb_stride = (int)(K / 8);
  if (mimo && cw == 0) {
    q->pg.tx_scheme = cfg->tx_scheme; q->pg.nof_tb = cfg->tbs2 ? 2 : 1;
    q->pg.codebook_idx = (int)(cfg->tbs2 ? cfg->pmi + 1 : cfg->pmi); // pdsch.c:914
    if (cfg->tbs2) {
      srslte_hip_dl_rx_cfg_t c1 = *cfg;
      c1.mod = cfg->mod2; c1.tbs = cfg->tbs2; c1.mod2 = 0; c1.tbs2 = 0;
      q->cw1 = dl_rx_create_impl(&c1, 1);
      if (!q->cw1) {
        srslte_hip_dl_rx_destroy(q);
        return nullptr;
      }
      q->pg.mod1 = c1.mod; q->pg.Qm1 = q->cw1->pg.Qm; q->pg.max_bits1 = q->cw1->pg.max_bits; q->pg.scr_words1 = q->cw1->pg.scr_words;
      q->pg.scr1 = q->cw1->d_scr; q->pg.csi1 = q->cw1->d_csi; q->pg.csi_max1 = q->cw1->d_csi_max;
    }
  }
  return q;
}

extern "C" srslte_hip_dl_rx_t* srslte_hip_dl_rx_create(const srslte_hip_dl_rx_cfg_t* cfg) { return dl_rx_create_impl(cfg, 0); }

extern "C" int srslte_hip_dl_rx_keep_symbols(srslte_hip_dl_rx_t* q, int enable)
{
  if (!q) return SRSLTE_ERROR_INVALID_INPUTS;
  if (q->cw1) {
    if (int r = srslte_hip_dl_rx_keep_symbols(q->cw1, enable)) return r;
  }
  if (enable && !q->d_d) {
    HIP_TRY(hipMalloc((void**)&q->d_d, sizeof(cf32) * (size_t)q->pg.max_re * q->cfg.max_batch));
  } else if (!enable && q->d_d) {
    HIP_TRY(hipFree(q->d_d));
    q->d_d = nullptr;
  }
  return SRSLTE_SUCCESS;
}

extern "C" uint32_t srslte_hip_dl_rx_nof_re(const srslte_hip_dl_rx_t* q, uint32_t sf_idx)
{
  return q ? (uint32_t)q->pg.cls[sf_idx % 10 == 0 ? 0 : (sf_idx % 10 == 5 ? 1 : 2)].nof_re : 0;
}

extern "C" const void* srslte_hip_dl_rx_debug_buffer(const srslte_hip_dl_rx_t* q, int which)
{
  if (!q) return nullptr;
  if (which >= 100) return srslte_hip_dl_rx_debug_buffer(q->cw1, which - 100); // the second codeword's buffers
  switch (which) {
    case 0: return q->d_grid;
    case 1: return q->d_ce;
    case 2: return q->d_res;
    case 3: return q->d_d;
    case 4: return q->d_e;
    case 5: return q->d_w;
    case 6: return q->d_cb_iters;
    case 7: return q->d_cb_ok;
    case 8: return q->d_cb_bytes;
    case 9: return q->d_csi;
    case 10: return q->d_csi_max;
    // grants mode (srslte_hip_dl_rx_batch_grants): 11 e [nof_sf][max_bits], 12 w [nof_sf * Cmax][stride], 13 cb iters, 14 cb ok,
    // 15 RE lists [nof_sf][max_re], 16 scrambling words [nof_sf][words]
    case 11: return q->gs ? q->gs->d_e : nullptr;
    case 12: return q->gs ? q->gs->d_w : nullptr;
    case 13: return q->gs ? q->gs->d_cb_iters : nullptr;
    case 14: return q->gs ? q->gs->d_cb_ok : nullptr;
    case 15: return q->gs ? q->gs->d_relist : nullptr;
    case 16: return q->gs ? q->gs->d_scr : nullptr;
    case 17: return q->gs ? q->gs->d_csi : nullptr;
  }
  return nullptr;
}

// Stage launchers, also used one by one by bench.py to time each kernel in isolation.
extern "C" int srslte_hip_dl_rx_stage(srslte_hip_dl_rx_t* q, int stage, const void* d_iq, uint32_t tti0, uint32_t nof_sf, uint8_t* d_tb,
                                      uint32_t tb_stride, uint8_t* d_tb_ok, void* stream)
{
  if (!q || nof_sf > q->cfg.max_batch) return SRSLTE_ERROR_INVALID_INPUTS;
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  hipStream_t    st = (hipStream_t)stream;
  const uint32_t C = q->seg.C, K = q->seg.K1;
  const cf32*    grid = q->grid_in ? q->grid_in : q->d_grid;
  switch (stage) {
    case 0: return srslte_hip_ofdm_rx_sf_batch(q->ofdm, d_iq, q->d_grid, (int)nof_sf * q->pg.nof_rx, stream); // [sf][rx] = nof_sf * nof_rx subframes
    case 1:
      return srslte_hip_chest_dl_estimate_batch_multi(q->chest, &q->cfg.chest_cfg, tti0, grid, q->d_ce, q->d_res, (int)nof_sf, q->pg.nof_rx, stream);
    case 2: {
      PdschGeom g = q->pg;
      g.tti0      = (int)tti0;
      if (g.csi_max) HIP_TRY(hipMemsetAsync(g.csi_max, 0, sizeof(uint32_t) * nof_sf, st));
      if (g.tx_scheme) {
        if (g.csi_max1) HIP_TRY(hipMemsetAsync(g.csi_max1, 0, sizeof(uint32_t) * nof_sf, st));
        hipLaunchKernelGGL(pdsch_demod_mimo_kernel<int16_t>, dim3(ceil_div(g.max_re, 256), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                           (const ChestResDev*)q->d_res, (const uint32_t*)q->d_scr, q->d_d, q->cw1 ? q->cw1->d_d : (cf32*)nullptr, q->d_e,
                           q->cw1 ? q->cw1->d_e : (int16_t*)nullptr, g);
      } else if (g.nof_ports == 4) {
        if (q->cfg.llr_8bit) {
          hipLaunchKernelGGL(pdsch_demod_div4_kernel<int8_t>, dim3(ceil_div(g.max_re, 1024), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                             (const uint32_t*)q->d_scr, q->d_d, (int8_t*)q->d_e, g);
        } else {
          hipLaunchKernelGGL(pdsch_demod_div4_kernel<int16_t>, dim3(ceil_div(g.max_re, 1024), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                             (const uint32_t*)q->d_scr, q->d_d, q->d_e, g);
        }
      } else if (g.nof_ports == 2) {
        if (q->cfg.llr_8bit) {
          hipLaunchKernelGGL(pdsch_demod_div_kernel<int8_t>, dim3(ceil_div(g.max_re, 512), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                             (const uint32_t*)q->d_scr, q->d_d, (int8_t*)q->d_e, g);
        } else {
          hipLaunchKernelGGL(pdsch_demod_div_kernel<int16_t>, dim3(ceil_div(g.max_re, 512), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                             (const uint32_t*)q->d_scr, q->d_d, q->d_e, g);
        }
      } else if (q->cfg.llr_8bit) {
        hipLaunchKernelGGL(pdsch_demod_kernel<int8_t>, dim3(ceil_div(g.max_re, 256), nof_sf), dim3(256), 0, st, grid,
                           (const cf32*)q->d_ce, (const ChestResDev*)q->d_res, (const uint32_t*)q->d_scr, q->d_d, (int8_t*)q->d_e, g);
      } else {
        hipLaunchKernelGGL(pdsch_demod_kernel<int16_t>, dim3(ceil_div(g.max_re, 256), nof_sf), dim3(256), 0, st, grid,
                           (const cf32*)q->d_ce, (const ChestResDev*)q->d_res, (const uint32_t*)q->d_scr, q->d_d, q->d_e, g);
      }
      LAUNCH_CHECK();
      return SRSLTE_SUCCESS;
    }
    case 3: {
      if (q->cw1) {
        if (int r = srslte_hip_dl_rx_stage(q->cw1, 3, nullptr, tti0, nof_sf, nullptr, 0, nullptr, stream)) return r;
      }
      RmGeom g = q->rg;
      g.tti0   = (int)tti0;
      g.combine = q->harq_combine;
      g.skip    = q->harq_combine ? q->d_cb_ok : nullptr;
      const uint32_t* tbl = q->d_rm_tbl_rv[q->harq_rv];
      if (q->cfg.llr_8bit) {
        if (rm_fits_lds(g, 1)) {
          hipLaunchKernelGGL(rm_rx_lds_kernel<int8_t>, dim3(nof_sf * C), dim3(256), rm_lds_bytes(g, 1), st, (const int8_t*)q->d_e, (int8_t*)q->d_w,
                             tbl, g);
        } else {
          hipLaunchKernelGGL(rm_rx_kernel<int8_t>, dim3(ceil_div(g.w_stride, 1024), nof_sf * C), dim3(256), 0, st, (const int8_t*)q->d_e,
                             (int8_t*)q->d_w, tbl, g);
        }
      } else {
        if (rm_fits_lds(g)) {
          hipLaunchKernelGGL(rm_rx_lds_kernel<int16_t>, dim3(nof_sf * C), dim3(256), rm_lds_bytes(g, 2), st, (const int16_t*)q->d_e, q->d_w, tbl, g);
        } else {
          hipLaunchKernelGGL(rm_rx_kernel<int16_t>, dim3(ceil_div(g.w_stride, 512), nof_sf * C), dim3(256), 0, st, (const int16_t*)q->d_e,
                             q->d_w, tbl, g);
        }
      }
      LAUNCH_CHECK();
      return SRSLTE_SUCCESS;
    }
    case 4:
      if (q->cw1) {
        if (int r = srslte_hip_dl_rx_stage(q->cw1, 4, nullptr, tti0, nof_sf, nullptr, 0, nullptr, stream)) return r;
      }
      tdec_set_tb_syndrome(q->tdec, q->d_tb_rem, C, q->d_cb_syn);
      tdec_set_skip(q->tdec, q->harq_combine ? q->d_cb_ok : nullptr);
      return tdec_run_batch_w(q->tdec, q->d_w, q->cfg.llr_8bit ? 1 : 0, q->in_stride, q->W != 0, K, -1, nof_sf * C, q->cfg.max_iterations,
                              C > 1 ? 0x1800063u : 0x1864CFBu, C > 1 ? K : q->cfg.tbs + 24, q->d_cb_bytes, K / 8, q->d_cb_iters, q->d_cb_ok, st);
    case 5: {
      if (!d_tb || !d_tb_ok || tb_stride < q->cfg.tbs / 8 + 6) return SRSLTE_ERROR_INVALID_INPUTS;
      if (q->cw1) { // rows nof_sf .. 2 nof_sf - 1 of d_tb / d_tb_ok: the second transport block of every subframe
        if (int r = srslte_hip_dl_rx_stage(q->cw1, 5, nullptr, tti0, nof_sf, d_tb + (size_t)nof_sf * tb_stride, tb_stride, d_tb_ok + nof_sf, stream)) return r;
      }
      TbGeom g    = q->tg;
      g.tb_stride = (int)tb_stride;
      if (q->d_tb_rem) {
        hipLaunchKernelGGL(tb_asm_kernel, dim3(nof_sf), dim3(256), 0, st, (const uint8_t*)q->d_cb_bytes, (const uint8_t*)q->d_cb_ok,
                           (const uint32_t*)q->d_cb_syn, d_tb, d_tb_ok, g);
      } else {
        hipLaunchKernelGGL(tb_crc_kernel, dim3(nof_sf), dim3(512), 0, st, (const uint8_t*)q->d_cb_bytes, (const uint8_t*)q->d_cb_ok,
                           (const uint32_t*)q->d_tbcrc, d_tb, d_tb_ok, g);
      }
      LAUNCH_CHECK();
      return SRSLTE_SUCCESS;
    }
  }
  return SRSLTE_ERROR_INVALID_INPUTS;
}

extern "C" int srslte_hip_dl_rx_batch(srslte_hip_dl_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, uint8_t* d_tb,
                                      uint32_t tb_stride, uint8_t* d_tb_ok, void* stream)
{
  if (!q || !d_iq || !d_tb || !d_tb_ok) return SRSLTE_ERROR_INVALID_INPUTS;
  for (int s = 0; s < 6; s++) {
    int r = srslte_hip_dl_rx_stage(q, s, d_iq, tti0, nof_sf, d_tb, tb_stride, d_tb_ok, stream);
    if (r) return r;
  }
  return SRSLTE_SUCCESS;
}

// HARQ (decode_tb_cb, sch.c:299-414, on a srslte_softbuffer_rx_t per transport block, softbuffer.c:46-150): slot b of the object keeps
// its code blocks' soft buffers, CRC flags and decoded bytes between calls. new_data != 0 starts new transport blocks (what the MAC's
// srslte_softbuffer_rx_reset_tbs does on a toggled NDI): buffers are overwritten, every block is decoded. new_data == 0 is a
// retransmission with redundancy version rv: the de-matched LLRs are ADDED to the kept soft buffers (rm_turbo.c:407-409), blocks whose
// CRC already passed are neither combined nor decoded again. srslte_hip_dl_rx_batch is rv 0 / new data.
extern "C" int srslte_hip_dl_rx_batch_harq(srslte_hip_dl_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, uint32_t rv, int new_data,
                                           uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream)
{
  if (!q || !d_iq || !d_tb || !d_tb_ok || rv > 3) return SRSLTE_ERROR_INVALID_INPUTS;
  const uint32_t rv2[2] = {rv, rv};
  const int      nd2[2] = {new_data, new_data};
  return srslte_hip_dl_rx_batch_harq2(q, d_iq, tti0, nof_sf, rv2, nd2, d_tb, tb_stride, d_tb_ok, stream);
}

// the same with a redundancy version and a new-data flag per transport block (two-layer modes: each block has its own HARQ process state,
// srslte_pdsch_cfg_t.softbuffers.rx[0 / 1] and grant.tb[0 / 1].rv)
extern "C" int srslte_hip_dl_rx_batch_harq2(srslte_hip_dl_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, const uint32_t rv[2],
                                            const int new_data[2], uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream)
{
  if (!q || !d_iq || !d_tb || !d_tb_ok || !rv || !new_data || rv[0] > 3 || rv[1] > 3) return SRSLTE_ERROR_INVALID_INPUTS;
  srslte_hip_dl_rx_t* objs[2] = {q, q->cw1};
  for (int c = 0; c < 2; c++) {
    srslte_hip_dl_rx_t* o = objs[c];
    if (!o) continue;
    if (!o->d_rm_tbl_rv[rv[c]]) {
      if (int r = dl_rx_rm_table(o, rv[c], &o->d_rm_tbl_rv[rv[c]])) return r;
    }
    o->harq_rv      = rv[c];
    o->harq_combine = new_data[c] ? 0 : 1;
  }
  int r = SRSLTE_SUCCESS;
  for (int s = 0; s < 6 && !r; s++) r = srslte_hip_dl_rx_stage(q, s, d_iq, tti0, nof_sf, d_tb, tb_stride, d_tb_ok, stream);
  for (srslte_hip_dl_rx_t* o : objs) {
    if (o) {
      o->harq_rv      = 0;
      o->harq_combine = 0;
    }
  }
  return r;
}

// Same chain from resource grids already in the frequency domain (what follows srslte_ofdm_rx_sf in ue_dl.c:375-397): stages 1..5
extern "C" int srslte_hip_dl_rx_grid_batch(srslte_hip_dl_rx_t* q, const void* d_grid, uint32_t tti0, uint32_t nof_sf, uint8_t* d_tb,
                                           uint32_t tb_stride, uint8_t* d_tb_ok, void* stream)
{
  if (!q || !d_grid || !d_tb || !d_tb_ok) return SRSLTE_ERROR_INVALID_INPUTS;
  q->grid_in = (const cf32*)d_grid;
  int r      = SRSLTE_SUCCESS;
  for (int s = 1; s < 6 && r == SRSLTE_SUCCESS; s++) r = srslte_hip_dl_rx_stage(q, s, nullptr, tti0, nof_sf, d_tb, tb_stride, d_tb_ok, stream);
  q->grid_in = nullptr;
  return r;
}

// --------------------------------------------------------------------------------------------------------------------
// Per-subframe grants (what a TTI stream looks like: srslte_pdsch_decode takes a new srslte_pdsch_grant_t every subframe, pdsch.c:833-997):
// subframe b of the batch is received with grants[b] - PRB masks of both slots (srslte_pdsch_grant_t.prb_idx, walked by srslte_pdsch_cp
// pdsch.c:81-206), modulation, transport block size, redundancy version, RNTI, CFI, new-data flag. RE lists and scrambling sequences are made
// on the device from the grants, rate de-matching runs over the ragged set of code blocks of the batch, the turbo decoder once per block
// length present in it. cfg.tbs bounds the transport block size (buffer sizes), cfg.mod / cfg.rnti / cfg.cfi are not used here.
// Single-port cells (TM1) and 2- / 4-port cells with transmit diversity (TM2), 1..4 receive antennas, 16- or 8-bit LLRs (cfg.llr_8bit), with or
// without the CSI weighting of cfg.csi_enable.
// --------------------------------------------------------------------------------------------------------------------
// Gold-sequence basis (sequence.c:48-79) for scr_gen_kernel: row 0 = the x1 sequence, row 1 + j = the x2 sequence of c_init = 1 << j, `words`
// packed words each; all 31 x2 basis sequences advance together, bit j of the state word = basis j
static int gold_basis_upload(uint32_t words, uint32_t** d_basis)
{
  const uint32_t         nbits = words * 32, Nc = 1600, tot = nbits + Nc + 31;
  std::vector<uint8_t>   x1(tot);
  std::vector<uint32_t>  x2(tot);
  for (uint32_t n = 0; n < 31; n++) {
    x1[n] = n == 0;
    x2[n] = 1u << n;
  }
  for (uint32_t n = 0; n + 31 < tot; n++) {
    x1[n + 31] = x1[n + 3] ^ x1[n];
    x2[n + 31] = x2[n + 3] ^ x2[n + 2] ^ x2[n + 1] ^ x2[n];
  }
  std::vector<uint32_t> basis((size_t)32 * words, 0);
  for (uint32_t n = 0; n < nbits; n++) {
    const uint32_t w = n >> 5, b = n & 31, v2 = x2[n + Nc];
    basis[w] |= (uint32_t)x1[n + Nc] << b;
    for (uint32_t j = 0; j < 31; j++) basis[(size_t)(1 + j) * words + w] |= ((v2 >> j) & 1u) << b;
  }
  return upload(d_basis, basis);
}

// Buffers of a grants mode: V per-transport-block slots of up to Cmax code blocks and max_re resource elements each; relist_rows > 0 adds the
// PDSCH RE lists, csi the CSI rows. Shared by the downlink (slot = subframe, or max_batch + subframe for codeword 1) and the uplink (slot = PUSCH).
static int grants_alloc(GrantsState* g, uint32_t max_re, uint32_t V, uint32_t Cmax, uint32_t relist_rows, bool csi, size_t extra_desc_bytes)
{
  g->Cmax     = Cmax;
  g->stride   = (srslte_hip_tdec_input_len(6144, 1) + 31) & ~31u;
  g->max_re   = max_re;                            // upper bound of any allocation
  g->max_bits = (g->max_re * 8 + 15) & ~15u;       // 256QAM
  g->words    = (g->max_re * 8 + 31) / 32 + 2;     // + the spare word the demapper reads
  g->V        = V;
  g->tdec     = srslte_hip_tdec_create(6144, g->V * g->Cmax);
  g->d_relist = g->d_scr = g->d_basis = g->d_cb_iters = nullptr;
  g->d_e = g->d_w = nullptr;
  g->d_cb_bytes = g->d_cb_ok = g->d_desc = nullptr;
  g->d_csi = nullptr; g->d_csi_max = nullptr;
  g->h_slot = 0;
  for (int i = 0; i < 4; i++) {
    g->h_pin[i]  = nullptr;
    g->h_used[i] = false;
  }
  if (!g->tdec) return SRSLTE_ERROR;
  if (gold_basis_upload(g->words, &g->d_basis)) return SRSLTE_ERROR;
  const size_t nblk = (size_t)g->V * g->Cmax;
  g->desc_bytes     = sizeof(GrantDev) * g->V + sizeof(SfDesc) * g->V + sizeof(CbDesc) * nblk + sizeof(uint32_t) * nblk + extra_desc_bytes;
  for (int i = 0; i < 4; i++) {
    HIP_TRY(hipEventCreateWithFlags(&g->h_ev[i], hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void**)&g->h_pin[i], g->desc_bytes));
  }
  if (relist_rows) HIP_TRY(hipMalloc((void**)&g->d_relist, sizeof(uint32_t) * (size_t)g->max_re * relist_rows));
  HIP_TRY(hipMalloc((void**)&g->d_scr, sizeof(uint32_t) * (size_t)g->words * g->V));
  HIP_TRY(hipMalloc((void**)&g->d_e, sizeof(int16_t) * ((size_t)g->max_bits * g->V + 16)));
  HIP_TRY(hipMalloc((void**)&g->d_w, sizeof(int16_t) * (size_t)g->stride * nblk));
  HIP_TRY(hipMalloc((void**)&g->d_cb_bytes, (size_t)768 * nblk));
  HIP_TRY(hipMalloc((void**)&g->d_cb_ok, nblk));
  HIP_TRY(hipMalloc((void**)&g->d_cb_iters, sizeof(uint32_t) * nblk));
  HIP_TRY(hipMalloc((void**)&g->d_desc, g->desc_bytes));
  if (csi) {
    HIP_TRY(hipMalloc((void**)&g->d_csi, sizeof(float) * (size_t)g->max_re * g->V));
    HIP_TRY(hipMalloc((void**)&g->d_csi_max, sizeof(uint32_t) * g->V));
  }
  // HARQ state of slots that have not seen new data yet: nothing decoded, empty soft buffers
  HIP_TRY(hipMemset(g->d_cb_ok, 0, nblk));
  HIP_TRY(hipMemset(g->d_w, 0, sizeof(int16_t) * (size_t)g->stride * nblk));
  HIP_TRY(hipMemset(g->d_cb_bytes, 0, (size_t)768 * nblk));
  HIP_TRY(hipDeviceSynchronize());
  return SRSLTE_SUCCESS;
}

static void grants_free(GrantsState* g)
{
  if (!g) return;
  srslte_hip_tdec_destroy(g->tdec);
  void* gb[] = {g->d_relist, g->d_scr, g->d_basis, g->d_cb_iters, g->d_e, g->d_w, g->d_cb_bytes, g->d_cb_ok, g->d_desc, g->d_csi, g->d_csi_max};
  for (void* b : gb) {
    if (b) (void)hipFree(b);
  }
  for (auto& kv : g->rm_tbl) (void)hipFree(kv.second);
  for (auto& kv : g->crc_fac) (void)hipFree(kv.second);
  for (int i = 0; i < 4; i++) {
    if (g->h_pin[i]) { // its event was created just before it
      (void)hipHostFree(g->h_pin[i]);
      (void)hipEventDestroy(g->h_ev[i]);
    }
  }
  delete g;
}

static int grants_init(srslte_hip_dl_rx_t* q)
{
  auto*          g = new GrantsState();
  const uint32_t P = q->cfg.nof_prb, B = q->cfg.max_batch;
  q->gs = g;
  return grants_alloc(g, 14 * 12 * P, (q->pg.nof_ports == 2 && q->pg.nof_rx == 2) ? 2 * B : B, q->seg.C, B, q->cfg.csi_enable != 0, 0);
}

// slot table of (K, rv) in the input layout of the decoder AUTO selects for K, stride = that layout's length rounded up to 32
static int grants_rm_table(GrantsState* g, uint32_t K, uint32_t rv, uint32_t W, uint32_t w_len, const uint32_t** d_tbl)
{
  auto it = g->rm_tbl.find({K, rv});
  if (it == g->rm_tbl.end()) {
    std::vector<uint32_t> t;
    lte_rm_rx_table(K, rv, t);
    if (W) {
      for (auto& v : t) v = v < 3 * K ? (v % 3) * (K + 32) + ((v / 3) % (K / W)) * W + (v / 3) / (K / W) : (v - 3 * K) + 3 * (K + 32);
    }
    uint32_t* d = nullptr;
    if (upload(&d, rm_slot_table(t, w_len))) return SRSLTE_ERROR;
    it = g->rm_tbl.emplace(std::make_pair(K, rv), d).first;
  }
  *d_tbl = it->second;
  return SRSLTE_SUCCESS;
}

// weights of tb_crc_bytes_kernel for a transport block of tbs bits: w[t] = x^(8 cB (255 - t)) mod g_CRC24A
static int grants_crc_factors(GrantsState* g, uint32_t tbs, const uint32_t** d_fac)
{
  auto it = g->crc_fac.find(tbs);
  if (it == g->crc_fac.end()) {
    const uint32_t poly = 0x1864CFBu, nbytes = tbs / 8 + 3, cB = (nbytes + 255) / 256;
    auto           mul  = [&](uint32_t a, uint32_t b) {
      uint32_t r = 0;
      for (int i = 23; i >= 0; i--) {
        r <<= 1;
        if (r & 0x1000000u) r ^= poly;
        if ((b >> i) & 1) r ^= a;
      }
      return r;
    };
    uint32_t m = 1; // x^(8 cB) mod g
    for (uint32_t i = 0; i < 8 * cB; i++) {
      m <<= 1;
      if (m & 0x1000000u) m ^= poly;
    }
    std::vector<uint32_t> w(256);
    w[255] = 1;
    for (int t = 254; t >= 0; t--) w[t] = mul(w[t + 1], m);
    uint32_t* d = nullptr;
    if (upload(&d, w)) return SRSLTE_ERROR;
    it = g->crc_fac.emplace(tbs, d).first;
  }
  *d_fac = it->second;
  return SRSLTE_SUCCESS;
}

// The PRB masks of a grant as the RE-list kernel wants them, upstream's stale `offset` values, and the number of PDSCH REs of the allocation
// (what pdsch_relist_kernel will list). gd.sf_idx / gd.lstart must be set; shared by the receive and the transmit grants modes.
static uint32_t pdsch_grant_dev(const srslte_hip_dl_grant_t& gr, uint32_t P, uint32_t cell_id, int npt, GrantDev& gd)
{
  const uint32_t sf_idx = (uint32_t)gd.sf_idx, lstart = (uint32_t)gd.lstart;
  bool any0 = false, below1 = false;
  for (uint32_t n = 0; n < P; n++) {
    for (int s_ = 0; s_ < 2; s_++) {
      if ((gr.prb_mask[s_][n >> 5] >> (n & 31)) & 1u) {
        gd.mask[s_][n >> 5] |= 1u << (n & 31);
        if (s_ == 0) any0 = true;
        if (s_ == 1 && n + 3 < P / 2) below1 = true;
      }
    }
  }
  // upstream's `offset` when it reaches the half PRBs of slot 1, symbol 0 (pdsch.c:147-157,:172-190): set by the whole PRBs before them;
  // with 2 / 4 ports every CRS symbol sets the same value
  bool any1_whole = false; // any whole (non-centre) PRB of slot 1: by symbol 1 its symbol-0 row has set `offset` too
  for (uint32_t n = 0; n < P; n++) {
    if (((gr.prb_mask[1][n >> 5] >> (n & 31)) & 1u) && !(n >= P / 2 - 3 && n < P / 2 + 3 + (P % 2))) any1_whole = true;
  }
  gd.q_off = npt == 1 ? (below1 ? (int)(cell_id % 6) : (any0 ? (int)((cell_id + 3) % 6) : 0))
                      : (((below1 || any0) ? (int)(cell_id % 3) : 0) | (((below1 || any0 || any1_whole) ? (int)(cell_id % 3) : 0) << 8));
  // number of PDSCH REs (what pdsch_relist_kernel will list; srslte_ra_dl_grant_nof_re): per symbol, whole PRBs carry 12 REs (10 with
  // CRS, 8 on a multi-port cell), PRBs inside the PSS / SSS / PBCH region of a sync symbol none, the two PRBs an odd bandwidth cuts in half
  // there half of that
  auto pop = [](const uint32_t* m, const uint32_t* f) {
    return __builtin_popcount(m[0] & f[0]) + __builtin_popcount(m[1] & f[1]) + __builtin_popcount(m[2] & f[2]) + __builtin_popcount(m[3] & f[3]);
  };
  uint32_t centre[4] = {0, 0, 0, 0}, half[4] = {0, 0, 0, 0}, all[4] = {~0u, ~0u, ~0u, ~0u};
  for (uint32_t n = P / 2 - 3; n < P / 2 + 3 + (P % 2); n++) centre[n >> 5] |= 1u << (n & 31);
  if (P % 2) {
    half[(P / 2 - 3) >> 5] |= 1u << ((P / 2 - 3) & 31);
    half[(P / 2 + 3) >> 5] |= 1u << ((P / 2 + 3) & 31);
  }
  uint32_t nre = 0;
  for (int sym = 0; sym < 14; sym++) {
    const int s_ = sym / 7, l = sym % 7;
    if (s_ == 0 && l < (int)lstart) continue;
    const bool ref = l == 0 || l == 4 || (l == 1 && npt == 4), sync = (s_ == 0 && (sf_idx == 0 || sf_idx == 5) && l >= 5) || (s_ == 1 && sf_idx == 0 && l < 4);
    const int  per = ref ? (npt == 1 ? 10 : 8) : 12;
    nre += per * pop(gd.mask[s_], all);
    if (sync) nre += (per / 2) * pop(gd.mask[s_], half) - per * pop(gd.mask[s_], centre);
  }
  return nre;
}

// Host side of one grants-mode call: transport blocks are added one by one (descriptor of their slot, one descriptor per code block, decoder
// group by block length), then grants_back_end runs rate de-matching, the decoders and the transport-block check over what was added.
struct GrantsBuild {
  struct Group { uint32_t K, single; std::vector<uint32_t> slots; };
  GrantsState*       g;
  SfDesc*            h_sf;
  CbDesc*            h_cb;
  bool               l8;      // 8-bit LLRs (pdsch.c:760-779, sch.c:336-356): same buffers, as bytes
  uint32_t           max_tbs; // the object's largest transport block (its buffers are sized for it)
  int                npt, max_mod;
  const char*        who;
  std::vector<Group> groups;
  uint32_t           ncb = 0, max_seg = 0;
  // one transport block into slot v: descriptor, code-block descriptors, decoder group. b: the caller's index, for the message
  int add_tb(uint32_t b, uint32_t v, int mod, uint32_t tbs, uint32_t rv, int new_data, uint32_t nre, uint32_t Nl, uint32_t e_off = 0)
  {
    SfDesc& sd = h_sf[v];
    srslte_hip_cbsegm_t seg;
    if (mod < 1 || mod > max_mod || rv > 3 || tbs > max_tbs || tbs > (uint32_t)TB_MAX_BITS || (tbs % 8) || srslte_hip_cbsegm(&seg, tbs) || seg.F || seg.C2 ||
        seg.C > g->Cmax) {
      hip_log("[srslte_hip] %s grants: entry %u: unsupported transport block (mod %d, tbs %u, rv %u)\n", who, b, mod, tbs, rv);
      return SRSLTE_ERROR_INVALID_INPUTS;
    }
    const uint32_t Qm = 2 * (uint32_t)mod, K = seg.K1, C = seg.C;
    if (nre == 0 || nre < C * Nl) {
      hip_log("[srslte_hip] %s grants: entry %u: empty allocation\n", who, b);
      return SRSLTE_ERROR_INVALID_INPUTS;
    }
    sd.nof_re = (int)nre; sd.mod = mod; sd.Qm = (int)Qm; sd.C = (int)C; sd.K = (int)K; sd.tbs = (int)tbs; sd.rlen = (int)(C == 1 ? K : K - 24);
    if (grants_crc_factors(g, tbs, &sd.crc_fac)) return SRSLTE_ERROR;
    const uint32_t W = l8 ? srslte_hip_tdec_autoimp_get_subblocks_8bit(K) : srslte_hip_tdec_autoimp_get_subblocks(K);
    const uint32_t w_len = (srslte_hip_tdec_input_len(K, W != 0) + 31) & ~31u;
    const uint32_t* tbl = nullptr;
    if (grants_rm_table(g, K, rv, W, w_len, &tbl)) return SRSLTE_ERROR;
    Group* grp = nullptr;
    for (auto& x : groups) {
      if (x.K == K && x.single == (C == 1 ? tbs : 0)) grp = &x;
    }
    if (!grp) {
      groups.push_back(Group{K, C == 1 ? tbs : 0, {}});
      grp = &groups.back();
    }
    for (uint32_t c = 0; c < C; c++) {
      CbDesc& cd = h_cb[ncb++];
      cd.sf = (int)v; cd.cb = (int)c; cd.C = (int)C; cd.K = (int)K; cd.Qm = (int)Qm; cd.nof_re = (int)nre; cd.combine = new_data ? 0 : 1;
      cd.w_len = (int)w_len; cd.tbl = tbl; cd.Nl = (int)Nl; cd.e_off = (int)e_off;
      grp->slots.push_back(v * g->Cmax + c);
    }
    const uint32_t seg_bytes = (Qm * (nre / C) + 2 * Qm * (uint32_t)npt) * (l8 ? 1 : 2) + 32;
    max_seg = seg_bytes > max_seg ? seg_bytes : max_seg;
    return SRSLTE_SUCCESS;
  }
  uint32_t fill_map(uint32_t* h_map) const
  {
    uint32_t n = 0;
    for (auto& x : groups) {
      for (uint32_t v : x.slots) h_map[n++] = v;
    }
    return n;
  }
};

// rate de-matching of every added block, the decoders group by group, and the transport-block check of rows 0 .. nrows-1 (row -> slot: the row
// itself, or cw1_off + row - nof_rows0 for rows behind the first nof_rows0: the second codewords)
static int grants_back_end(GrantsState* g, const GrantsBuild& bd, const SfDesc* d_sf, const CbDesc* d_cb, const uint32_t* d_map, uint32_t tti0,
                           uint32_t max_iterations, uint32_t nrows, uint32_t nof_rows0, uint32_t cw1_off, uint8_t* d_tb, uint32_t tb_stride,
                           uint8_t* d_tb_ok, hipStream_t st)
{
  const bool l8 = bd.l8;
  if (bd.ncb) {
    RmGeom rg;
    memset(&rg, 0, sizeof(rg));
    rg.cbd = d_cb; rg.cb_ok_rst = g->d_cb_ok; rg.C = (int)g->Cmax; rg.tti0 = (int)tti0; rg.max_bits = (int)g->max_bits; rg.w_stride = (int)g->stride;
    rg.Nl = 1; rg.skip = g->d_cb_ok; rg.max_re = (int)g->max_re; rg.csi = g->d_csi; rg.csi_max = g->d_csi_max;
    const int lds = (int)((bd.max_seg + 15) & ~15u);
    const uint32_t ncb = bd.ncb;
    if (l8 && lds <= 64 * 1024) {
      hipLaunchKernelGGL(rm_rx_lds_kernel<int8_t>, dim3(ncb), dim3(256), lds, st, (const int8_t*)g->d_e, (int8_t*)g->d_w, (const uint32_t*)nullptr, rg);
    } else if (l8) {
      hipLaunchKernelGGL(rm_rx_kernel<int8_t>, dim3(ceil_div((int)g->stride, 1024), ncb), dim3(256), 0, st, (const int8_t*)g->d_e, (int8_t*)g->d_w,
                         (const uint32_t*)nullptr, rg);
    } else if (lds <= 64 * 1024) {
      hipLaunchKernelGGL(rm_rx_lds_kernel<int16_t>, dim3(ncb), dim3(256), lds, st, (const int16_t*)g->d_e, g->d_w, (const uint32_t*)nullptr, rg);
    } else {
      hipLaunchKernelGGL(rm_rx_kernel<int16_t>, dim3(ceil_div((int)g->stride, 512), ncb), dim3(256), 0, st, (const int16_t*)g->d_e, g->d_w,
                         (const uint32_t*)nullptr, rg);
    }
    LAUNCH_CHECK();
    tdec_set_tb_syndrome(g->tdec, nullptr, 1, nullptr);
    tdec_set_skip(g->tdec, g->d_cb_ok);
    uint32_t off = 0;
    for (auto& x : bd.groups) {
      const uint32_t n = (uint32_t)x.slots.size();
      const uint32_t W = l8 ? srslte_hip_tdec_autoimp_get_subblocks_8bit(x.K) : srslte_hip_tdec_autoimp_get_subblocks(x.K);
      tdec_set_cb_map(g->tdec, d_map + off);
      const int r = tdec_run_batch_w(g->tdec, g->d_w, l8 ? 1 : 0, g->stride, W != 0, x.K, -1, n, max_iterations, x.single ? 0x1864CFBu : 0x1800063u,
                                     x.single ? x.single + 24 : x.K, g->d_cb_bytes, 768, g->d_cb_iters, g->d_cb_ok, st);
      tdec_set_cb_map(g->tdec, nullptr);
      if (r) return r;
      off += n;
    }
  }
  TbGeom tg;
  memset(&tg, 0, sizeof(tg));
  tg.desc = d_sf; tg.C = (int)g->Cmax; tg.cb_stride = 768; tg.tb_stride = (int)tb_stride; tg.nof_sf = (int)nof_rows0; tg.cw1_off = (int)cw1_off;
  hipLaunchKernelGGL(tb_crc_bytes_kernel, dim3(nrows), dim3(256), 0, st, (const uint8_t*)g->d_cb_bytes, (const uint8_t*)g->d_cb_ok, d_tb, d_tb_ok, tg);
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}

static int grants_run(srslte_hip_dl_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, const srslte_hip_dl_grant2_t* grants, uint8_t* d_tb,
                      uint32_t tb_stride, uint8_t* d_tb_ok, void* stream, bool second_rows);

extern "C" int srslte_hip_dl_rx_batch_grants(srslte_hip_dl_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, const srslte_hip_dl_grant_t* grants,
                                             uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream)
{
  if (!q || !grants || nof_sf > q->cfg.max_batch) return SRSLTE_ERROR_INVALID_INPUTS;
  std::vector<srslte_hip_dl_grant2_t> g2(nof_sf);
  for (uint32_t b = 0; b < nof_sf; b++) {
    memset(&g2[b], 0, sizeof(g2[b]));
    g2[b].tb0 = grants[b];
  }
  return grants_run(q, d_iq, tti0, nof_sf, g2.data(), d_tb, tb_stride, d_tb_ok, stream, false);
}

// The same with the transmission scheme per subframe and a second transport block: on a 2-port cell received with 2 antennas a grant can be
// transmit diversity (tx_scheme 0 / 1: DCI 1 / 1A), large-delay CDD (3, two transport blocks) or closed-loop multiplexing (2, two blocks
// with pmi 0-1 or one with pmi 0-3), as srslte_ra_dl_dci_to_grant makes them (ra_dl.c:530-600). Codeword 1 of subframe b is a second
// per-subframe slot (descriptor, LLR row, CSI row, HARQ soft buffers) max_batch further on; its transport block is row nof_sf + b of d_tb.
extern "C" int srslte_hip_dl_rx_batch_grants2(srslte_hip_dl_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, const srslte_hip_dl_grant2_t* grants,
                                              uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream)
{
  return grants_run(q, d_iq, tti0, nof_sf, grants, d_tb, tb_stride, d_tb_ok, stream, true);
}

// The same from frequency-domain grids (what follows srslte_ofdm_rx_sf in ue_dl.c:375-397; d_grid as srslte_hip_dl_rx_grid_batch takes it)
extern "C" int srslte_hip_dl_rx_grid_batch_grants2(srslte_hip_dl_rx_t* q, const void* d_grid, uint32_t tti0, uint32_t nof_sf, const srslte_hip_dl_grant2_t* grants,
                                                   uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream)
{
  if (!q || !d_grid) return SRSLTE_ERROR_INVALID_INPUTS;
  q->grid_in  = (const cf32*)d_grid;
  const int r = grants_run(q, d_grid, tti0, nof_sf, grants, d_tb, tb_stride, d_tb_ok, stream, true);
  q->grid_in  = nullptr;
  return r;
}

// second_rows: rows nof_sf .. 2 nof_sf - 1 of d_tb / d_tb_ok exist (srslte_hip_dl_rx_batch_grants2 on a cell where two-layer grants can occur)
static int grants_run(srslte_hip_dl_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, const srslte_hip_dl_grant2_t* grants, uint8_t* d_tb,
                      uint32_t tb_stride, uint8_t* d_tb_ok, void* stream, bool second_rows)
{
  if (!q || !d_iq || !grants || !d_tb || !d_tb_ok || nof_sf > q->cfg.max_batch || tb_stride < q->cfg.tbs / 8 + 6) return SRSLTE_ERROR_INVALID_INPUTS;
  if (q->cfg.tx_scheme) {
    hip_log("[srslte_hip] dl_rx grants mode: create the object without a fixed two-layer scheme; the grants carry it\n");
    return SRSLTE_ERROR;
  }
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  if (!q->gs && grants_init(q)) { // a failed start leaves no half-made state behind
    grants_free(q->gs);
    q->gs = nullptr;
    return SRSLTE_ERROR;
  }
  GrantsState*   g  = q->gs;
  hipStream_t    st = (hipStream_t)stream;
  const uint32_t P = q->cfg.nof_prb, cell_id = q->cfg.cell_id, B = q->cfg.max_batch, V = g->V;
  const int      npt = q->pg.nof_ports;
  const size_t   nblk = (size_t)V * g->Cmax;
  const uint32_t hs = g->h_slot++ & 3u;
  if (g->h_used[hs]) HIP_TRY(hipEventSynchronize(g->h_ev[hs])); // the copy that last read this buffer (four calls ago) has completed
  auto*          h_gr = reinterpret_cast<GrantDev*>(g->h_pin[hs]);
  auto*          h_sf = reinterpret_cast<SfDesc*>(h_gr + V);
  auto*          h_cb = reinterpret_cast<CbDesc*>(h_sf + V);
  auto*          h_map = reinterpret_cast<uint32_t*>(h_cb + nblk);
  auto*          d_gr = reinterpret_cast<GrantDev*>(g->d_desc);
  auto*          d_sf = reinterpret_cast<SfDesc*>(d_gr + V);
  auto*          d_cb = reinterpret_cast<CbDesc*>(d_sf + V);
  auto*          d_map = reinterpret_cast<uint32_t*>(d_cb + nblk);
  const bool         l8 = q->cfg.llr_8bit != 0; // the 8-bit LLR path the applications select (pdsch.c:760-779, sch.c:336-356): same buffers, as bytes
  bool               any_mimo = false;
  GrantsBuild        bd;
  bd.g = g; bd.h_sf = h_sf; bd.h_cb = h_cb; bd.l8 = l8; bd.max_tbs = q->cfg.tbs; bd.npt = npt; bd.max_mod = 4; bd.who = "dl_rx";
  auto add_tb = [&](uint32_t b, uint32_t v, int mod, uint32_t tbs, uint32_t rv, int new_data, uint32_t nre, uint32_t Nl) -> int {
    return bd.add_tb(b, v, mod, tbs, rv, new_data, nre, Nl);
  };
  for (uint32_t b = 0; b < nof_sf; b++) {
    const srslte_hip_dl_grant2_t& g2 = grants[b];
    const srslte_hip_dl_grant_t&  gr = g2.tb0;
    GrantDev&                     gd = h_gr[b];
    SfDesc&                       sd = h_sf[b];
    memset(&gd, 0, sizeof(gd));
    memset(&sd, 0, sizeof(sd));
    if (V > B) {
      memset(&h_gr[B + b], 0, sizeof(gd));
      memset(&h_sf[B + b], 0, sizeof(sd));
    }
    const uint32_t sf_idx = (tti0 + b) % 10, lstart = gr.cfi + (P < 10 ? 1 : 0);
    gd.sf_idx = (int)sf_idx; gd.lstart = (int)lstart; gd.rnti = gr.rnti;
    sd.idx = g->d_relist + (size_t)b * g->max_re;
    sd.scr = g->d_scr + (size_t)b * g->words;
    if (gr.tbs == 0) continue; // no transport block in this subframe: C = 0, tb_ok = 0
    const bool two_layer = g2.tx_scheme >= 2;
    if (gr.cfi < 1 || gr.cfi > 3 || g2.tx_scheme < 0 || g2.tx_scheme > 3 ||
        (two_layer && (V == B || (g2.tx_scheme == 3 && g2.tbs2 == 0) || (g2.tbs2 ? g2.pmi > 1 : g2.pmi > 3))) || (!two_layer && g2.tbs2)) {
      hip_log("[srslte_hip] dl_rx grants: subframe %u: unsupported grant (cfi %u, tx_scheme %d, pmi %u, second transport block %u bits)\n", b, gr.cfi, g2.tx_scheme,
              g2.pmi, g2.tbs2);
      return SRSLTE_ERROR_INVALID_INPUTS;
    }
    const uint32_t nre = pdsch_grant_dev(gr, P, cell_id, npt, gd);
    if (!two_layer && (nre % (uint32_t)npt)) { // the transmit-diversity pre-decoders take the REs in groups of nof_ports (precoding.c:564-650)
      hip_log("[srslte_hip] dl_rx grants: subframe %u: %u REs are not a multiple of the %d ports\n", b, nre, npt);
      return SRSLTE_ERROR_INVALID_INPUTS;
    }
    // nof_layers == nof_tb for the two-layer modes: N_L = 1; transmit diversity: 2 (srslte_dlsch_decode2, sch.c:507-531)
    if (int r = add_tb(b, b, gr.mod, gr.tbs, gr.rv, gr.new_data, nre, (!two_layer && npt > 1) ? 2 : 1)) return r;
    sd.scheme = g2.tx_scheme; sd.nof_tb = g2.tbs2 ? 2 : 1; sd.codebook = (int)(g2.tbs2 ? g2.pmi + 1 : g2.pmi); // pdsch.c:914
    if (two_layer) any_mimo = true;
    if (g2.tbs2) { // codeword 1: slot max_batch + b, the same REs, its own sequence (q = 1), modulation, transport block
      GrantDev& gd1 = h_gr[B + b];
      SfDesc&   sd1 = h_sf[B + b];
      gd1 = gd; gd1.cw = 1;
      sd1.idx = sd.idx;
      sd1.scr = g->d_scr + (size_t)(B + b) * g->words;
      if (int r = add_tb(b, B + b, g2.mod2, g2.tbs2, g2.rv2, g2.new_data2, nre, 1)) return r;
      sd1.scheme = g2.tx_scheme; sd1.nof_tb = 2; sd1.codebook = sd.codebook;
    }
  }
  bd.fill_map(h_map);
  // stages 0, 1: OFDM demodulation and channel estimation do not depend on the grants
  int r = q->grid_in ? SRSLTE_SUCCESS : srslte_hip_dl_rx_stage(q, 0, d_iq, tti0, nof_sf, d_tb, tb_stride, d_tb_ok, stream); // grids from the caller: no OFDM stage
  if (!r) r = srslte_hip_dl_rx_stage(q, 1, d_iq, tti0, nof_sf, d_tb, tb_stride, d_tb_ok, stream);
  if (r) return r;
  // the descriptors
  HIP_TRY(hipMemcpyAsync(g->d_desc, g->h_pin[hs], g->desc_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(g->h_ev[hs], st));
  g->h_used[hs] = true;
  const uint32_t nrows = (second_rows && V > B) ? 2 * nof_sf : nof_sf; // transport-block rows of the call
  hipLaunchKernelGGL(pdsch_relist_kernel, dim3(nof_sf), dim3(RELIST_THREADS), 0, st, (const GrantDev*)d_gr, g->d_relist, (int)P, (int)cell_id, (int)g->max_re,
                     q->pg.nof_ports);
  hipLaunchKernelGGL(scr_gen_kernel, dim3(ceil_div((int)g->words, 256), nof_sf), dim3(256), 0, st, (const GrantDev*)d_gr, (const uint32_t*)g->d_basis, g->d_scr,
                     (int)g->words, (int)cell_id);
  if (any_mimo) { // the sequences of the second codewords
    hipLaunchKernelGGL(scr_gen_kernel, dim3(ceil_div((int)g->words, 256), nof_sf), dim3(256), 0, st, (const GrantDev*)(d_gr + B), (const uint32_t*)g->d_basis,
                       g->d_scr + (size_t)B * g->words, (int)g->words, (int)cell_id);
  }
  LAUNCH_CHECK();
  {
    PdschGeom pg = q->pg;
    pg.desc = d_sf; pg.tti0 = (int)tti0; pg.max_re = (int)g->max_re; pg.max_bits = (int)g->max_bits; pg.csi = g->d_csi; pg.csi_max = g->d_csi_max;
    if (g->d_csi_max) HIP_TRY(hipMemsetAsync(g->d_csi_max, 0, sizeof(uint32_t) * V, st));
    const cf32* grid = q->grid_in ? q->grid_in : q->d_grid;
    if (pg.nof_ports == 4) {
      if (l8) {
        hipLaunchKernelGGL(pdsch_demod_div4_kernel<int8_t>, dim3(ceil_div((int)g->max_re, 1024), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                           (const uint32_t*)g->d_scr, (cf32*)nullptr, (int8_t*)g->d_e, pg);
      } else {
        hipLaunchKernelGGL(pdsch_demod_div4_kernel<int16_t>, dim3(ceil_div((int)g->max_re, 1024), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                           (const uint32_t*)g->d_scr, (cf32*)nullptr, g->d_e, pg);
      }
    } else if (pg.nof_ports == 2) {
      if (l8) {
        hipLaunchKernelGGL(pdsch_demod_div_kernel<int8_t>, dim3(ceil_div((int)g->max_re, 512), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                           (const uint32_t*)g->d_scr, (cf32*)nullptr, (int8_t*)g->d_e, pg);
      } else {
        hipLaunchKernelGGL(pdsch_demod_div_kernel<int16_t>, dim3(ceil_div((int)g->max_re, 512), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                           (const uint32_t*)g->d_scr, (cf32*)nullptr, g->d_e, pg);
      }
      if (any_mimo) { // the two-layer subframes (the kernels above skipped them); codeword 1's rows sit max_batch further on in every buffer
        pg.cw1_off  = (int)B;
        pg.max_bits1 = pg.max_bits;
        pg.csi1     = g->d_csi ? g->d_csi + (size_t)B * g->max_re : nullptr;
        pg.csi_max1 = g->d_csi_max ? g->d_csi_max + B : nullptr;
        if (l8) {
          hipLaunchKernelGGL(pdsch_demod_mimo_kernel<int8_t>, dim3(ceil_div((int)g->max_re, 256), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                             (const ChestResDev*)q->d_res, (const uint32_t*)g->d_scr, (cf32*)nullptr, (cf32*)nullptr, (int8_t*)g->d_e,
                             (int8_t*)g->d_e + (size_t)B * g->max_bits, pg);
        } else {
          hipLaunchKernelGGL(pdsch_demod_mimo_kernel<int16_t>, dim3(ceil_div((int)g->max_re, 256), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                             (const ChestResDev*)q->d_res, (const uint32_t*)g->d_scr, (cf32*)nullptr, (cf32*)nullptr, g->d_e, g->d_e + (size_t)B * g->max_bits, pg);
        }
      }
    } else if (l8) {
      hipLaunchKernelGGL(pdsch_demod_kernel<int8_t>, dim3(ceil_div((int)g->max_re, 256), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                         (const ChestResDev*)q->d_res, (const uint32_t*)g->d_scr, (cf32*)nullptr, (int8_t*)g->d_e, pg);
    } else {
      hipLaunchKernelGGL(pdsch_demod_kernel<int16_t>, dim3(ceil_div((int)g->max_re, 256), nof_sf), dim3(256), 0, st, grid, (const cf32*)q->d_ce,
                         (const ChestResDev*)q->d_res, (const uint32_t*)g->d_scr, (cf32*)nullptr, g->d_e, pg);
    }
    LAUNCH_CHECK();
  }
  return grants_back_end(g, bd, d_sf, d_cb, d_map, tti0, q->cfg.max_iterations, nrows, nof_sf, B, d_tb, tb_stride, d_tb_ok, st);
}

// ====================================================================================================================
// PUSCH receive pipeline (eNB side): OFDM RX with the -1/2 carrier shift (enb_ul.c:58-63) -> chest_ul ->
// RE extraction + one-tap MMSE (pusch.c:461-475) -> inverse transform precoding (:477-478) -> soft demap + descramble + UL
// channel de-interleaver (:480-503, sch.c:891-913) -> rate de-matching -> turbo decode -> TB CRC (decode_tb, sch.c:429-500).
// UL-SCH data only (no UCI multiplexing), same allocation in both slots, normal CP, not shortened, rv 0.
// ====================================================================================================================
namespace {

// HARQ-ACK on the PUSCH (36.212 5.2.2.6-5.2.2.8; srslte_uci_encode_ack_ri / _decode_ack_ri, uci.c:497-520,:547-602,:627-656,:695-788; 1 or 2 bits,
// no RI / CQI): ACK symbol i sits on data symbol {2,3,8,9}[(3i)%4] ({1,2,6,7} with at most 10 data symbols), sub-carrier M_sc-1-i/4.
struct AckGeom {
  int O, Qprime; // 0: no ACK
};
__device__ __forceinline__ int ack_symbol_index(const AckGeom& a, int n, int k, int M_sc, int nsymb)
{ // index i of the ACK symbol at data symbol n, sub-carrier k, or -1
  int colidx;
  if (nsymb > 10) {
    colidx = n == 2 ? 0 : (n == 3 ? 1 : (n == 8 ? 2 : (n == 9 ? 3 : -1)));
  } else {
    colidx = n == 1 ? 0 : (n == 2 ? 1 : (n == 6 ? 2 : (n == 7 ? 3 : -1)));
  }
  if (a.O == 0 || colidx < 0) return -1;
  const int i = 4 * (M_sc - 1 - k) + (3 * colidx) % 4; // (3 i) % 4 == colidx <=> i % 4 == (3 colidx) % 4
  return i < a.Qprime ? i : -1;
}
// rank indication: the same rule on the columns {1,4,7,10} ({0,3,5,8} with at most 10 symbols; uci.c:521-545)
__device__ __forceinline__ int ri_symbol_index(const AckGeom& a, int n, int k, int M_sc, int nsymb)
{
  int colidx;
  if (nsymb > 10) {
    colidx = n == 1 ? 0 : (n == 4 ? 1 : (n == 7 ? 2 : (n == 10 ? 3 : -1)));
  } else {
    colidx = n == 0 ? 0 : (n == 3 ? 1 : (n == 5 ? 2 : (n == 8 ? 3 : -1)));
  }
  if (a.O == 0 || colidx < 0) return -1;
  const int i = 4 * (M_sc - 1 - k) + (3 * colidx) % 4;
  return i < a.Qprime ? i : -1;
}
// number of RI symbols before (k, n) in the row-by-row order the channel interleaver fills (ulsch_interleave_gen, sch.c:580-598):
// the UL-SCH symbol at (k, n) is symbol k nsymb + n - ri_before of the rate-matched stream
__device__ __forceinline__ int ri_before(const AckGeom& a, int n, int k, int M_sc, int nsymb)
{
  if (a.O == 0) return 0;
  int cnt = max(0, a.Qprime - 4 * (M_sc - k)); // rows above k in the matrix are sub-carriers below it: RI symbols i >= 4 (M_sc - k)
  const int i0 = 4 * (M_sc - 1 - k);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c   = (3 * j) % 4;
    const int col = nsymb > 10 ? 1 + 3 * c : (c == 0 ? 0 : (c == 1 ? 3 : (c == 2 ? 5 : 8)));
    cnt += (i0 + j < a.Qprime && col < n) ? 1 : 0;
  }
  return cnt;
}
__device__ __forceinline__ int ack_bit_type(const uint8_t* ack, int O, int Qm, int e)
{ // encode_ri_ack (uci.c:573-602) repeated: 0 / 1 value, 2 repetition of the previous bit, 3 placeholder
  if (O == 1) {
    const int r = e % Qm;
    return r == 0 ? ack[0] : (r == 1 ? 2 : 3);
  }
  const int r = e % (3 * Qm), s3 = r / Qm, b = r % Qm;
  if (b >= 2) return 3;
  const int v = (2 * s3 + b) % 3; // o0 o1 | o2 o0 | o1 o2
  return v == 0 ? ack[0] : (v == 1 ? ack[1] : (ack[0] ^ ack[1]));
}

struct PuschGeom {
  int cell_nre, M_sc, n_prb, n_prb1, mod, Qm, tti0, scr_words, mmse; // n_prb / n_prb1: PRB offset of slot 0 / slot 1 (grant.n_prb_tilde[2])
  AckGeom ack, ri;
  int*    ack_sum; // [nof_sf][4] accumulators of the ACK decisions (zeroed per call), or null
  int*    ri_sum;  // the same for the rank indication
  int nsymb; // data symbols per subframe: 12, or 11 when the last symbol is left to the SRS (shortened subframe, pusch.c:52-91)
};

__device__ __forceinline__ int pusch_data_symbol(int n) { return n < 3 ? n : (n < 9 ? n + 1 : n + 2); } // 12 data symbols skip l = 3, 10

// grid = (ceil(M_sc/256), 12, nof_sf): z[sf][n][k] = y h* / (|h|^2 + noise) on the granted PRBs of data symbol n
__global__ __launch_bounds__(256) void pusch_eq_kernel(const cf32* __restrict__ grid, const cf32* __restrict__ ce,
                                                       const float* __restrict__ noise /* stride 5 floats */, cf32* __restrict__ z, PuschGeom g)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y, sf = blockIdx.z;
  if (k >= g.M_sc) return;
  const int    l  = pusch_data_symbol(n);
  const size_t o  = ((size_t)sf * 14 + l) * g.cell_nre + (l < 7 ? g.n_prb : g.n_prb1) * 12 + k;
  const cf32   y = grid[o], h = ce[o];
  const float  n0 = g.mmse ? noise[sf * 5] : 0.f;
  const float  re = y.x * h.x + y.y * h.y, im = y.y * h.x - y.x * h.y, csi = h.x * h.x + h.y * h.y + n0; // precoding.c:277-288, scaling 1
  z[((size_t)sf * g.nsymb + n) * g.M_sc + k] = make_float2(re * 1.0f / csi, im * 1.0f / csi);
}

// grid = (ceil(M_sc/64), nof_sf), 256 threads: demap the 12 symbols of 64 sub-carriers, descramble in the received (symbol-major)
// bit order and store at the de-interleaved position g[(k * 12 + n) * Qm + b] = q[(n * M_sc + k) * Qm + b] (36.212 5.2.2.8 without
// UCI). The 64 x 12 x Qm LLRs of a workgroup are contiguous in g: they are collected in LDS and leave with 16-byte stores
// (a 2-byte store per lane at a 24 Qm byte stride costs the L1 one cache line per lane).
// d: the PUSCH's de-precoded symbols [nsymb][M_sc]; cs: its scrambling bits; gout: its LLR row; sf: its row of the UCI accumulators
__device__ __forceinline__ void pusch_demod_body(const cf32* __restrict__ d, const uint32_t* __restrict__ cs, int16_t* __restrict__ gout, const PuschGeom& g,
                                                 int sf, int k0, int16_t* stage)
{
  const int nsym = g.nsymb * g.M_sc;
  const int nk = min(64, g.M_sc - k0);
  const int rb0 = ri_before(g.ri, 0, k0, g.M_sc, g.nsymb); // RI symbols on the sub-carriers before this workgroup's
  for (int t = threadIdx.x; t < 64 * g.nsymb; t += 256) {
    const int n = t >> 6, kl = t & 63;
    if (kl >= nk) continue;
    const int i = n * g.M_sc + k0 + kl;
    short     o[8];
    demod_dev::demod_s(g.mod, d[i], i, nsym, o);
    const int      bit0 = i * g.Qm;
    const uint32_t c2   = (uint32_t)((((uint64_t)cs[(bit0 >> 5) + 1] << 32) | cs[bit0 >> 5]) >> (bit0 & 31));
    const int      ai   = ack_symbol_index(g.ack, n, k0 + kl, g.M_sc, g.nsymb);
    const int      ri   = ri_symbol_index(g.ri, n, k0 + kl, g.M_sc, g.nsymb);
    if (ri >= 0) { // the interleaver left this symbol out of the UL-SCH stream (sch.c:968-979): it only feeds the RI decision
      const int d0 = (c2 & 1) ? -o[0] : o[0], d1 = (c2 & 2) ? -o[1] : o[1];
      if (g.ri.O == 1) {
        atomicAdd(&g.ri_sum[sf * 4], d0 + ((c2 & 1) ? -o[1] : o[1]));
      } else if (3 * (ri / 3) + 3 < g.ri.Qprime) {
        const int s3 = ri % 3;
        atomicAdd(&g.ri_sum[sf * 4 + (s3 == 0 ? 0 : (s3 == 1 ? 2 : 1))], d0);
        atomicAdd(&g.ri_sum[sf * 4 + (s3 == 0 ? 1 : (s3 == 1 ? 0 : 2))], d1);
      }
      continue;
    }
    const int sl = kl * g.nsymb + n - (ri_before(g.ri, n, k0 + kl, g.M_sc, g.nsymb) - rb0); // symbol slot in this workgroup's stage
    if (ai >= 0) { // uci_decode_ri_ack (sch.c:929-966): this symbol feeds the ACK decision and reaches the decoder as zeros
      const int d0 = (c2 & 1) ? -o[0] : o[0], d1 = (c2 & 2) ? -o[1] : o[1]; // descrambled
      if (g.ack.O == 1) { // value bit + its repetition, which carries the value bit's scrambling (uci.c:627-640)
        atomicAdd(&g.ack_sum[sf * 4], d0 + ((c2 & 1) ? -o[1] : o[1]));
      } else if (3 * (ai / 3) + 3 < g.ack.Qprime) { // a triplet is only used if another symbol follows it (uci.c:776-777)
        const int s3 = ai % 3; // o0 o1 | o2 o0 | o1 o2
        atomicAdd(&g.ack_sum[sf * 4 + (s3 == 0 ? 0 : (s3 == 1 ? 2 : 1))], d0);
        atomicAdd(&g.ack_sum[sf * 4 + (s3 == 0 ? 1 : (s3 == 1 ? 0 : 2))], d1);
      }
      for (int b = 0; b < g.Qm; b++) stage[sl * g.Qm + b] = 0;
      continue;
    }
    for (int b = 0; b < g.Qm; b++) stage[sl * g.Qm + b] = ((c2 >> b) & 1) ? (short)-o[b] : o[b];
  }
  __syncthreads();
  if (g.ri.O && k0 == 0 && threadIdx.x == 0) {
    // ulsch_deinterleave scatters g[lut[i]] = q[i] in ascending i with lut = 0 at the RI positions (sch.c:589-591,:910): g[0] ends up
    // with the LLR of the LAST RI bit position: last bit of RI symbol 1 (symbol 0 if it is the only one), on the bottom sub-carrier; the
    // 1-bit decoder has re-scrambled its repetition bit by then (uci.c:632-633)
    const int n = g.ri.Qprime >= 2 ? (g.nsymb > 10 ? 10 : 8) : (g.nsymb > 10 ? 1 : 0), i = n * g.M_sc + g.M_sc - 1, bit = i * g.Qm + g.Qm - 1;
    short     o[8];
    demod_dev::demod_s(g.mod, d[i], i, nsym, o);
    const int cbit = (cs[bit >> 5] >> (bit & 31)) & 1;
    stage[0]       = (g.ri.O == 1 && g.Qm == 2) ? o[1] : (cbit ? (short)-o[g.Qm - 1] : o[g.Qm - 1]);
    __threadfence_block();
  }
  if (g.ri.O && k0 == 0) __syncthreads();
  const int    nsl    = nk * g.nsymb - (ri_before(g.ri, 0, k0 + nk, g.M_sc, g.nsymb) - rb0); // UL-SCH symbols of this workgroup
  const size_t first  = (size_t)(k0 * g.nsymb - rb0) * g.Qm;
  if (g.ri.O == 0 || (rb0 == 0 && nsl == nk * g.nsymb)) {
    const int    nbytes = nk * g.nsymb * g.Qm * 2; // nk is a multiple of 4, Qm even: a multiple of 16
    char*        dst    = reinterpret_cast<char*>(gout + first);
    const char*  src    = reinterpret_cast<const char*>(stage);
    for (int o16 = threadIdx.x * 16; o16 < nbytes; o16 += 256 * 16) *reinterpret_cast<uint4*>(dst + o16) = *reinterpret_cast<const uint4*>(src + o16);
  } else { // the few workgroups that hold RI symbols: neither the length nor the start is 16-byte granular any more
    for (int e = threadIdx.x; e < nsl * g.Qm; e += 256) gout[first + e] = stage[e];
  }
}


__global__ __launch_bounds__(256) void pusch_demod_kernel(const cf32* __restrict__ d, const uint32_t* __restrict__ scr, int16_t* __restrict__ gout,
                                                          PuschGeom g)
{
  __shared__ __attribute__((aligned(16))) int16_t stage[64 * 12 * 8];
  const int sf = blockIdx.y, sf_idx = (g.tti0 + sf) % 10, nsym = g.nsymb * g.M_sc;
  pusch_demod_body(d + (size_t)sf * nsym, scr + (size_t)sf_idx * g.scr_words /* one spare word behind every sequence */, gout + (size_t)sf * nsym * g.Qm, g, sf,
                   blockIdx.x * 64, stage);
}

// Per-PUSCH grants (srslte_hip_ul_rx_batch_grants): PUSCH p of a call has its own allocation, modulation and sequence; its symbols sit at zoff in
// the z / d buffers (PUSCHs of one L_prb next to each other, for the batched transform de-precoding), its LLRs in row p.
struct PuschDesc {
  int sf;                        // subframe of the batch whose grid it is in
  int M_sc, n_prb, n_prb1, mod, Qm;
  int zoff;                      // in cf32
  AckGeom ack, ri;
  int cqi_Q, cqi_O;              // LLRs and bits of its CQI report (0: none)
};

// grid = (ceil(max M_sc / 256), nsymb, nof_pusch)
__global__ __launch_bounds__(256) void pusch_eq_grants_kernel(const cf32* __restrict__ grid, const cf32* __restrict__ ce,
                                                              const float* __restrict__ noise /* stride 5 floats */, cf32* __restrict__ z,
                                                              const PuschDesc* __restrict__ desc, int cell_nre, int nsymb, int mmse)
{
  const PuschDesc& pd = desc[blockIdx.z];
  const int        k = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y, M_sc = pd.M_sc;
  if (k >= M_sc) return;
  const int    l  = pusch_data_symbol(n);
  const size_t o  = ((size_t)pd.sf * 14 + l) * cell_nre + (l < 7 ? pd.n_prb : pd.n_prb1) * 12 + k;
  const cf32   y = grid[o], h = ce[o];
  const float  n0 = mmse ? noise[blockIdx.z * 5] : 0.f;
  const float  re = y.x * h.x + y.y * h.y, im = y.y * h.x - y.x * h.y, csi = h.x * h.x + h.y * h.y + n0; // precoding.c:277-288, scaling 1
  z[(size_t)pd.zoff + (size_t)n * M_sc + k] = make_float2(re * 1.0f / csi, im * 1.0f / csi);
}

// grid = (ceil(max M_sc / 64), nof_pusch)
__global__ __launch_bounds__(256) void pusch_demod_grants_kernel(const cf32* __restrict__ d, const uint32_t* __restrict__ scr, int scr_words,
                                                                 int16_t* __restrict__ gout, int max_bits, const PuschDesc* __restrict__ desc, int nsymb,
                                                                 int* __restrict__ ack_sum, int* __restrict__ ri_sum)
{
  __shared__ __attribute__((aligned(16))) int16_t stage[64 * 12 * 8];
  const int p = blockIdx.y, k0 = blockIdx.x * 64;
  if (k0 >= desc[p].M_sc) return;
  PuschGeom g; // a local, not the kernel argument: filled from the descriptor
  g.cell_nre = 0; g.M_sc = desc[p].M_sc; g.n_prb = desc[p].n_prb; g.n_prb1 = desc[p].n_prb1; g.mod = desc[p].mod; g.Qm = desc[p].Qm; g.tti0 = 0;
  g.scr_words = scr_words; g.mmse = 0; g.ack = desc[p].ack; g.ri = desc[p].ri; g.ack_sum = ack_sum; g.ri_sum = ri_sum; g.nsymb = nsymb;
  pusch_demod_body(d + desc[p].zoff, scr + (size_t)p * scr_words, gout + (size_t)p * max_bits, g, p, k0, stage);
}

} // namespace

namespace {
// ---- CQI / PMI report on the PUSCH (36.212 5.2.2.6, 5.2.2.6.4; srslte_uci_decode_cqi_pusch, uci.c:423-467): the first Q = Q' Qm LLRs of the
// de-interleaved stream. Up to 11 bits: (32, O) block code - the repetitions are added up with wrapping int16 (srslte_vec_sum_sss) and the
// sum is correlated with all 2^O code words, the first word of the highest correlation wins (decode_cqi_short :305-341). Above 11 bits:
// srslte_rm_conv_rx_s (rm_conv.c:160-219, sequential because of its 10000 = "no value yet" sentinel), then the tail-biting K = 7 rate-1/3
// Viterbi decoder of viterbi37_avx2_16bit.c on three repetitions of the frame - one lane per state, predecessors by shuffle, decisions by
// ballot - with the soft bits scaled as srslte_viterbi_decode_f scales them (the reference's int16 wrapper overflows, see oracle/orc_cqi.c),
// and the CRC-8. One workgroup per subframe; the Viterbi runs on its first wavefront.
__constant__ uint16_t CQI_M32[32] = {0x403, 0x607, 0x749, 0x50D, 0x48F, 0x5D3, 0x755, 0x599, 0x69B, 0x65D, 0x6E5, 0x567, 0x7A9, 0x6AB, 0x4B1, 0x6F3,
                                     0x277, 0x139, 0x0FB, 0x061, 0x445, 0x60B, 0x591, 0x717, 0x3DF, 0x4E3, 0x32D, 0x3AF, 0x175, 0x1FD, 0x7FF, 0x001}; // Table 5.2.2.6.4-1
__constant__ uint8_t  CQI_PERM[32]     = {1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31, 0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30};
__constant__ uint8_t  CQI_PERM_INV[32] = {16, 0, 24, 8, 20, 4, 28, 12, 18, 2, 26, 10, 22, 6, 30, 14, 17, 1, 25, 9, 21, 5, 29, 13, 19, 3, 27, 11, 23, 7, 31, 15};
__device__ __forceinline__ uint32_t cqi_crc8(const uint8_t* bits, int n)
{
  uint32_t r = 0;
  for (int i = 0; i < n + 8; i++) {
    r = (r << 1) | (i < n ? (bits[i] & 1u) : 0u);
    if (r & 0x100u) r ^= 0x19Bu;
  }
  return r & 0xffu;
}

__global__ __launch_bounds__(256) void pusch_cqi_decode_kernel(const int16_t* __restrict__ gl, int g_stride, int Q_all, int O_all, uint8_t* __restrict__ cqi_out,
                                                               uint8_t* __restrict__ ok_out, const PuschDesc* __restrict__ desc)
{
  // per-PUSCH grants: the report's size comes from the row's descriptor (rows without a report are left alone)
  const int Q = desc ? desc[blockIdx.x].cqi_Q : Q_all, O = desc ? desc[blockIdx.x].cqi_O : O_all;
  if (O == 0) return;
  __shared__ int16_t            acc[32];
  __shared__ long long          best[4];
  __shared__ int16_t            tmp[3 * 96], dem[3 * 72];
  __shared__ uint16_t           us[3 * 72];
  __shared__ unsigned long long dec[3 * 72 + 6];
  __shared__ uint8_t            bits[3 * 72];
  const int      sf = blockIdx.x, tid = threadIdx.x;
  const int16_t* q  = gl + (size_t)sf * g_stride;
  uint8_t*       out = cqi_out + (size_t)sf * 64;
  if (O <= 11) {
    if (tid < 32) {
      int a = 0;
      for (int i = tid; i < Q; i += 32) a += q[i];
      acc[tid] = (int16_t)a; // wrapping, like the int16 adds of srslte_vec_sum_sss
    }
    __syncthreads();
    const int n = Q < 32 ? Q : 32;
    long long key = INT64_MIN;
    for (uint32_t w = tid; w < (1u << O); w += 256) {
      int corr = 0;
      for (int i = 0; i < n; i++) {
        uint32_t m = 0; // code bit i of word w: bit k of the report is bit O-1-k of w
        for (int k = 0; k < O; k++) m ^= ((w >> (O - 1 - k)) & 1u) & ((CQI_M32[i] >> k) & 1u);
        corr += m ? acc[i] : -acc[i];
      }
      const long long kk = ((long long)corr << 32) | (long long)(0xffffffffu - w); // highest correlation, then lowest word
      key = kk > key ? kk : key;
    }
    for (int o = 32; o > 0; o >>= 1) {
      const long long other = __shfl_xor(key, o, 64);
      key = other > key ? other : key;
    }
    if ((tid & 63) == 0) best[tid >> 6] = key;
    __syncthreads();
    if (tid == 0) {
      for (int i = 1; i < 4; i++) key = best[i] > key ? best[i] : key;
      const uint32_t w = 0xffffffffu - (uint32_t)(key & 0xffffffffll);
      for (int k = 0; k < O; k++) out[k] = (w >> (O - 1 - k)) & 1u;
      ok_out[sf] = 1;
    }
    return;
  }
  const int F = O + 8, nrows = (F - 1) / 32 + 1, K_p = nrows * 32, ndummy = K_p - F;
  for (int i = tid; i < 3 * K_p; i += 256) tmp[i] = 10000; // SRSLTE_RX_NULL
  for (int i = tid; i < 3 * F + 6; i += 256) dec[i] = 0ull;
  __syncthreads();
  if (tid == 0) {
    int k = 0, j = 0;
    while (k < Q) {
      const int d_i = (j % K_p) / nrows, d_j = (j % K_p) % nrows;
      if (d_j * 32 + CQI_PERM[d_i] >= ndummy) {
        const int16_t v = q[k];
        if (tmp[j] == 10000) {
          tmp[j] = v;
        } else if (v != 10000) {
          tmp[j] = (int16_t)(tmp[j] + v);
        }
        k++;
      }
      if (++j == 3 * K_p) j = 0;
    }
  }
  __syncthreads();
  for (int e = tid; e < 3 * F; e += 256) {
    const int     i = e / 3, s = e - 3 * i, d_i = (i + ndummy) / 32, d_j = (i + ndummy) % 32;
    const int16_t o = tmp[K_p * s + CQI_PERM_INV[d_j] * nrows + d_i];
    dem[e]          = o != 10000 ? o : (int16_t)0;
  }
  __syncthreads();
  if (tid >= 64) return;
  // srslte_viterbi_decode_f's quantisation (viterbi.c:532-540, srslte_vec_quant_fus): gain 1000 / max |.|, offset 32767.5, clip to 16 bits
  float mx = -9e9f;
  for (int i = tid; i < 3 * F; i += 64) mx = fmaxf(mx, fabsf((float)dem[i]));
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const float gain = 1000.0f / mx;
  for (int i = tid; i < 3 * F; i += 64) {
    const long t = (long)fmaf(gain, (float)dem[i], 32767.5f);
    us[i]        = (uint16_t)(t < 0 ? 0 : (t > 65535 ? 65535 : t));
  }
  __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): one wavefront from here on, LDS in order
  const int      n = tid, b = n >> 1;
  const uint32_t bt0 = (__builtin_popcount((2 * b) & 0x6D) & 1) ? 65535u : 0u, bt1 = (__builtin_popcount((2 * b) & 0x4F) & 1) ? 65535u : 0u,
                 bt2 = (__builtin_popcount((2 * b) & 0x57) & 1) ? 65535u : 0u;
  uint32_t old = 63;
  for (int t = 0; t < 3 * F; t++) {
    const int      f  = t % F;
    const uint32_t a = bt0 ^ us[3 * f], bb = bt1 ^ us[3 * f + 1], c = bt2 ^ us[3 * f + 2];
    const uint32_t m01 = (a + bb + 1) >> 1, met = ((c + m01 + 1) >> 1) >> 3, mm = 8191u - met;
    const uint32_t oi = (uint32_t)__shfl((int)old, b, 64), oj = (uint32_t)__shfl((int)old, b + 32, 64);
    const uint16_t x  = (uint16_t)(oi + ((n & 1) ? mm : met)), y = (uint16_t)(oj + ((n & 1) ? met : mm)); // (m0, m1) or (m2, m3)
    const bool     d  = (int16_t)(uint16_t)(x - y) > 0;
    old               = d ? y : x;
    const unsigned long long bal = __ballot(d);
    if (tid == 0) dec[t] = bal;
  }
  uint32_t mn = old;
  for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
  const unsigned long long at_min = __ballot(old == mn);
  if (tid == 0) {
    uint32_t endstate = (uint32_t)(63 - __builtin_clzll(at_min)) << 2; // the LAST state with the smallest metric
    for (int i = 3 * F - 1; i >= F; i--) {
      const uint32_t k = (uint32_t)(dec[6 + i] >> (endstate >> 2)) & 1u;
      endstate         = (endstate >> 1) | (k << 7);
      bits[i]          = (uint8_t)k;
    }
    const uint8_t* msg = bits + F; // the middle repetition
    uint32_t       rx  = 0;
    for (int i = 0; i < 8; i++) rx = (rx << 1) | msg[O + i];
    const bool good = cqi_crc8(msg, O) == rx;
    if (good) {
      for (int k = 0; k < O; k++) out[k] = msg[k];
    }
    ok_out[sf] = good ? 1 : 0;
  }
}
} // namespace

static int pusch_cqi_qprime(uint32_t O, uint32_t I_offset_cqi, uint32_t L_prb, uint32_t nsymb, uint32_t K_segm, uint32_t Qp_ri)
{ // Q_prime_cqi (uci.c:264-281), float arithmetic; 0 without a report
  static const float beta_cqi[16] = {-1.0f, -1.0f, 1.125f, 1.25f, 1.375f, 1.625f, 1.750f, 2.0f, 2.25f, 2.5f, 2.875f, 3.125f, 3.5f, 4.0f, 5.0f, 6.25f}; // sch.c:51-52
  if (O == 0) return 0;
  if (O > 64 || I_offset_cqi > 15 || beta_cqi[I_offset_cqi] < 0 || K_segm == 0) return -1;
  const uint32_t L = O < 11 ? 0 : 8;
  const uint32_t x = (uint32_t)ceilf((float)(O + L) * L_prb * 12 * nsymb * beta_cqi[I_offset_cqi] / K_segm), m = L_prb * 12 * nsymb - Qp_ri;
  return (int)(x < m ? x : m);
}

// Q' of the HARQ-ACK (Q_prime_ri_ack, uci.c:547-571, UL-SCH present): min(ceil(O M_sc N_symb beta / sum K_r), 4 M_sc) in float arithmetic
static int pusch_ack_qprime(uint32_t O, uint32_t I_offset_ack, uint32_t L_prb, uint32_t nsymb, uint32_t K_segm, bool is_ri = false)
{
  static const float beta_harq[16] = {2.0f, 2.5f, 3.125f, 4.0f, 5.0f, 6.250f, 8.0f, 10.0f, 12.625f, 15.875f, 20.0f, 31.0f, 50.0f, 80.0f, 126.0f, -1.0f}; // 36.213 Table 8.6.3-1
  static const float beta_ri[16] = {1.25f, 1.625f, 2.0f, 2.5f, 3.125f, 4.0f, 5.0f, 6.25f, 8.0f, 10.0f, 12.625f, 15.875f, 20.0f, -1.0f, -1.0f, -1.0f}; // Table 8.6.3-2 (sch.c:47-48)
  const float* beta = is_ri ? beta_ri : beta_harq;
  if (O == 0) return 0;
  if (O > 2 || I_offset_ack > 15 || beta[I_offset_ack] < 0 || K_segm == 0) return -1;
  const uint32_t x = (uint32_t)ceilf((float)O * L_prb * 12 * nsymb * beta[I_offset_ack] / K_segm), m = 4 * L_prb * 12;
  const uint32_t Qp = x < m ? x : m;
  return (Qp + 3) / 4 <= 12 * L_prb ? (int)Qp : -1; // the ACK rows must exist (uci.c:505)
}

namespace {
__global__ void pusch_ack_decide_kernel(const int* __restrict__ sum, uint8_t* __restrict__ ack, int nof_sf)
{
  const int sf = blockIdx.x * blockDim.x + threadIdx.x;
  if (sf >= nof_sf) return;
  ack[2 * sf]     = sum[4 * sf] > 0;     // uci.c:782-785
  ack[2 * sf + 1] = sum[4 * sf + 1] > 0;
}
} // namespace

struct srslte_hip_ul_rx {
  srslte_hip_ul_rx_cfg_t cfg;
  srslte_hip_ofdm_t*     ofdm;
  srslte_hip_chest_ul_t* chest;
  srslte_hip_tdec_t*     tdec;
  srslte_hip_cbsegm_t    seg;
  PuschGeom              pg;
  RmGeom                 rg;
  TbGeom                 tg;
  uint32_t               W, in_stride;
  uint32_t *             d_scr, *d_rm_tbl, *d_tbcrc, *d_tb_rem, *d_cb_syn, *d_cb_iters;
  uint32_t*              d_rm_tbl_rv[4]; // rate de-matching tables of redundancy versions 1-3, made on first use ([0] unused: d_rm_tbl)
  cf32 *                 d_grid, *d_ce, *d_z, *d_d;
  float*                 d_res; // [B] x srslte_hip_chest_ul_res_t
  int16_t *              d_g, *d_w;
  uint8_t *              d_cb_bytes, *d_cb_ok;
  int*                   d_ack_sum; // [B][4] ACK, then [B][4] RI
  uint8_t*               d_ack;     // [B][2] HARQ-ACK decisions of the last call, then [B][2] rank indications
  uint8_t*               d_cqi;     // [B][64] CQI report bits of the last call, then [B] CRC flags
  int                    Qp_cqi;
  // srslte_hip_ul_rx_batch_grants (created on first use): the shared grants machinery with one slot per PUSCH, plus the PUSCH front end's buffers
  struct GrantsState*    gs;
  cf32 *                 g_z, *g_d; // [max_grants][nsymb * 12 * nof_prb]
  float*                 g_res;     // [max_grants] x srslte_hip_chest_ul_res_t
  int*                   g_uci_sum; // [max_grants][4] HARQ-ACK accumulators, then the same for the rank indication
  uint8_t*               g_uci;     // [max_grants][2] HARQ-ACK decisions of the last grants call, then [max_grants][2] rank indications
  uint8_t*               g_cqi;     // [max_grants][64] CQI report bits of the last grants call, then [max_grants] CRC flags
};

extern "C" const uint8_t* srslte_hip_ul_rx_ack(const srslte_hip_ul_rx_t* q) { return q ? q->d_ack : nullptr; }
extern "C" const uint8_t* srslte_hip_ul_rx_ri(const srslte_hip_ul_rx_t* q) { return q ? q->d_ack + 2 * q->cfg.max_batch : nullptr; }
extern "C" const uint8_t* srslte_hip_ul_rx_cqi(const srslte_hip_ul_rx_t* q) { return q ? q->d_cqi : nullptr; }
extern "C" const uint8_t* srslte_hip_ul_rx_grants_ack(const srslte_hip_ul_rx_t* q) { return q ? q->g_uci : nullptr; }
extern "C" const uint8_t* srslte_hip_ul_rx_grants_cqi(const srslte_hip_ul_rx_t* q) { return q ? q->g_cqi : nullptr; }
extern "C" const uint8_t* srslte_hip_ul_rx_grants_ri(const srslte_hip_ul_rx_t* q)
{
  return q && q->g_uci ? q->g_uci + 2 * (q->cfg.max_grants ? q->cfg.max_grants : q->cfg.max_batch) : nullptr;
}

extern "C" void srslte_hip_ul_rx_destroy(srslte_hip_ul_rx_t* q)
{
  if (!q) return;
  srslte_hip_ofdm_destroy(q->ofdm);
  srslte_hip_chest_ul_destroy(q->chest);
  srslte_hip_tdec_destroy(q->tdec);
  void* bufs[] = {q->d_scr, q->d_rm_tbl, q->d_tbcrc, q->d_tb_rem, q->d_cb_syn, q->d_cb_iters, q->d_grid, q->d_ce, q->d_z,
                  q->d_d,   q->d_res,    q->d_g,     q->d_w,      q->d_cb_bytes, q->d_cb_ok, q->d_ack_sum, q->d_ack, q->d_cqi,
                  q->d_rm_tbl_rv[1], q->d_rm_tbl_rv[2], q->d_rm_tbl_rv[3], q->g_z, q->g_d, q->g_res, q->g_uci_sum, q->g_uci, q->g_cqi};
  for (void* b : bufs) {
    if (b) (void)hipFree(b);
  }
  grants_free(q->gs);
  delete q;
}

extern "C" srslte_hip_ul_rx_t* srslte_hip_ul_rx_create(const srslte_hip_ul_rx_cfg_t* cfg)
{
  if (!cfg || cfg->max_batch == 0 || cfg->mod < 1 || cfg->mod > 3 || cfg->max_iterations == 0 || cfg->L_prb < 1 ||
      cfg->n_prb + cfg->L_prb > cfg->nof_prb || (cfg->hopping && cfg->n_prb_slot1 + cfg->L_prb > cfg->nof_prb) ||
      !srslte_hip_dft_precoding_valid_prb(cfg->L_prb)) {
    hip_log("[srslte_hip] ul_rx: invalid configuration\n");
    return nullptr;
  }
  auto* q = new srslte_hip_ul_rx();
  memset(q, 0, sizeof(*q));
  q->cfg = *cfg;
  if (srslte_hip_cbsegm(&q->seg, cfg->tbs) || q->seg.F || q->seg.C2 || (cfg->tbs % 8)) {
    hip_log("[srslte_hip] ul_rx: TBS %u needs filler bits or two code-block sizes; not supported on device yet\n", cfg->tbs);
    delete q;
    return nullptr;
  }
  const uint32_t P = cfg->nof_prb, B = cfg->max_batch, C = q->seg.C, K = q->seg.K1, Qm = 2 * (uint32_t)cfg->mod, M_sc = 12 * cfg->L_prb;
  const uint32_t nsymb = cfg->shortened ? 11 : 12; // 2 (7 - 1) - N_srs data symbols (pusch.c:335-343)
  const uint32_t nof_re = nsymb * M_sc, nbits = nof_re * Qm, scr_words = (nbits + 31) / 32 + 1; // spare word: the demapper reads two per symbol
  const int      Qp_ri = pusch_ack_qprime(cfg->ri_len, cfg->I_offset_ri, cfg->L_prb, nsymb, C * K, true);
  const int      Qp_cqi = Qp_ri >= 0 ? pusch_cqi_qprime(cfg->cqi_len, cfg->I_offset_cqi, cfg->L_prb, nsymb, C * K, (uint32_t)Qp_ri) : -1;
  q->ofdm  = srslte_hip_ofdm_create((int)P, 1, 1);
  q->chest = srslte_hip_chest_ul_create(cfg->cell_id, P, 1, &cfg->dmrs_cfg);
  q->tdec  = srslte_hip_tdec_create(K, B * C);
  bool ok  = q->ofdm && q->chest && q->tdec && srslte_hip_ofdm_set_freq_shift(q->ofdm, -0.5f) == SRSLTE_SUCCESS; // enb_ul.c:62-63 (no normalisation)
  if (ok) { // srslte_sequence_pusch (sequences.c:65-67), one sequence per subframe index
    std::vector<uint32_t> scr((size_t)10 * scr_words, 0);
    std::vector<uint8_t>  c;
    for (uint32_t sf = 0; sf < 10; sf++) {
      lte_gold_sequence(((uint32_t)cfg->rnti << 14) + (sf << 9) + cfg->cell_id, nbits, c);
      for (uint32_t i = 0; i < nbits; i++) scr[(size_t)sf * scr_words + (i >> 5)] |= (uint32_t)c[i] << (i & 31);
    }
    ok = upload(&q->d_scr, scr) == SRSLTE_SUCCESS;
  }
  q->W         = srslte_hip_tdec_autoimp_get_subblocks(K);
  q->in_stride = (srslte_hip_tdec_input_len(K, q->W != 0) + 31) & ~31u;
  if (ok) { // rate-dematching table in the decoder's input layout (rm_turbo.c:160-260), as for the PDSCH
    std::vector<uint32_t> t;
    lte_rm_rx_table(K, 0, t);
    if (q->W) {
      for (auto& v : t) {
        v = v < 3 * K ? (v % 3) * (K + 32) + ((v / 3) % (K / q->W)) * q->W + (v / 3) / (K / q->W) : (v - 3 * K) + 3 * (K + 32);
      }
    }
    ok = upload(&q->d_rm_tbl, rm_slot_table(t, q->in_stride)) == SRSLTE_SUCCESS;
  }
  if (ok) {
    std::vector<uint32_t> rem(cfg->tbs + 24);
    uint32_t              v = 1;
    for (int j = (int)cfg->tbs + 23; j >= 0; j--) {
      rem[j] = v;
      v <<= 1;
      if (v & 0x1000000) v ^= 0x1864CFB;
    }
    ok = upload(&q->d_tbcrc, rem) == SRSLTE_SUCCESS;
    if (ok && q->W) {
      const uint32_t        rlen = C == 1 ? K : K - 24, Lw = K / q->W;
      std::vector<uint32_t> t((size_t)C * K, 0);
      for (uint32_t c = 0; c < C; c++) {
        for (uint32_t n = 0; n < rlen; n++) {
          const uint32_t pos = c * rlen + n;
          if (pos < cfg->tbs + 24) t[(size_t)c * K + (n % Lw) * q->W + n / Lw] = rem[pos];
        }
      }
      ok = upload(&q->d_tb_rem, t) == SRSLTE_SUCCESS && hipMalloc((void**)&q->d_cb_syn, sizeof(uint32_t) * B * C) == hipSuccess;
    }
  }
  const size_t glen = (size_t)14 * 12 * P;
  ok = ok && hipMalloc((void**)&q->d_grid, sizeof(cf32) * glen * B) == hipSuccess &&
       hipMalloc((void**)&q->d_ce, sizeof(cf32) * glen * B) == hipSuccess && hipMemset(q->d_ce, 0, sizeof(cf32) * glen * B) == hipSuccess &&
       hipMalloc((void**)&q->d_z, sizeof(cf32) * (size_t)nof_re * B) == hipSuccess &&
       hipMalloc((void**)&q->d_d, sizeof(cf32) * (size_t)nof_re * B) == hipSuccess &&
       hipMalloc((void**)&q->d_res, sizeof(float) * 5 * B) == hipSuccess &&
       hipMalloc((void**)&q->d_g, sizeof(int16_t) * ((size_t)nbits * B + 16)) == hipSuccess &&
       hipMalloc((void**)&q->d_w, sizeof(int16_t) * (size_t)q->in_stride * B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_cb_bytes, (size_t)(K / 8) * B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_cb_ok, (size_t)B * C) == hipSuccess && hipMemset(q->d_cb_ok, 0, (size_t)B * C) == hipSuccess &&
       hipMemset(q->d_w, 0, sizeof(int16_t) * (size_t)q->in_stride * B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_cb_iters, sizeof(uint32_t) * B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_ack_sum, sizeof(int) * 8 * B) == hipSuccess && hipMalloc((void**)&q->d_ack, (size_t)4 * B) == hipSuccess &&
       hipMemset(q->d_ack, 0, (size_t)4 * B) == hipSuccess && pusch_ack_qprime(cfg->ack_len, cfg->I_offset_ack, cfg->L_prb, nsymb, C * K) >= 0 &&
       Qp_ri >= 0 && (uint32_t)Qp_ri < nof_re && Qp_cqi >= 0 && (uint32_t)(Qp_ri + Qp_cqi) + C < nof_re &&
       hipMalloc((void**)&q->d_cqi, (size_t)65 * B) == hipSuccess && hipMemset(q->d_cqi, 0, (size_t)65 * B) == hipSuccess;
  ok = ok && hipDeviceSynchronize() == hipSuccess; // the memsets above ran on the null stream
  if (!ok) {
    hip_log("[srslte_hip] ul_rx: initialisation failed\n");
    srslte_hip_ul_rx_destroy(q);
    return nullptr;
  }
  q->pg.cell_nre = 12 * (int)P; q->pg.M_sc = (int)M_sc; q->pg.n_prb = (int)cfg->n_prb; q->pg.n_prb1 = (int)(cfg->hopping ? cfg->n_prb_slot1 : cfg->n_prb); q->pg.mod = cfg->mod; q->pg.Qm = (int)Qm;
  q->pg.scr_words = (int)scr_words; q->pg.mmse = cfg->mmse; q->pg.nsymb = (int)nsymb;
  q->pg.ack.O = (int)cfg->ack_len; q->pg.ack.Qprime = pusch_ack_qprime(cfg->ack_len, cfg->I_offset_ack, cfg->L_prb, nsymb, C * K);
  q->pg.ack_sum = q->d_ack_sum;
  q->pg.ri.O = (int)cfg->ri_len; q->pg.ri.Qprime = Qp_ri; q->pg.ri_sum = q->d_ack_sum + 4 * B;
  q->rg.C = (int)C; q->rg.K = (int)K; q->rg.Qm = (int)Qm; q->rg.max_bits = (int)nbits; q->rg.w_stride = (int)q->in_stride; q->rg.Nl = 1;
  q->rg.out_len = (int)(3 * K + 12);
  // the UL-SCH is rate-matched to what the RI and the CQI report leave (sch.c:1157-1160) and follows the report in the stream
  q->rg.nof_re[0] = q->rg.nof_re[1] = q->rg.nof_re[2] = (int)nof_re - Qp_ri - Qp_cqi;
  q->rg.e_off = Qp_cqi * (int)Qm;
  q->Qp_cqi   = Qp_cqi;
  q->tg.C = (int)C; q->tg.K = (int)K; q->tg.tbs = (int)cfg->tbs; q->tg.rlen = (int)(C == 1 ? K : K - 24); q->tg.cb_stride = (int)(K / 8);
  return q;
}

extern "C" const void* srslte_hip_ul_rx_debug_buffer(const srslte_hip_ul_rx_t* q, int which)
{
  if (!q) return nullptr;
  switch (which) {
    case 0: return q->d_grid;
    case 1: return q->d_ce;
    case 2: return q->d_res;
    case 3: return q->d_d;
    case 4: return q->d_g;
    case 5: return q->d_w;
    case 6: return q->d_cb_iters;
    case 7: return q->d_cb_ok;
    case 8: return q->d_cb_bytes;
    case 9: return q->d_z;
    case 10: return q->d_ack;
    // per-PUSCH grants mode: estimator results, de-precoded symbols (at each PUSCH's offset), LLR rows, pass counts per block slot
    case 20: return q->g_res;
    case 21: return q->g_d;
    case 22: return q->gs ? q->gs->d_e : nullptr;
    case 23: return q->gs ? q->gs->d_cb_iters : nullptr;
  }
  return nullptr;
}

// d_iq: [nof_sf][15*N]; outputs as srslte_hip_dl_rx_batch. Subframe b is TTI tti0 + b.
extern "C" int srslte_hip_ul_rx_batch(srslte_hip_ul_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, uint8_t* d_tb, uint32_t tb_stride,
                                      uint8_t* d_tb_ok, void* stream)
{
  return srslte_hip_ul_rx_batch_harq(q, d_iq, tti0, nof_sf, 0, 1, d_tb, tb_stride, d_tb_ok, stream);
}

// HARQ on the uplink (srslte_ulsch_decode hands grant.tb.rv and cfg->softbuffers.rx to the same decode_tb as the downlink, sch.c:1063):
// slot b keeps its code blocks' soft buffers, CRC flags and bytes between calls, exactly as srslte_hip_dl_rx_batch_harq. new_data != 0
// starts new transport blocks; new_data == 0 adds the de-matched LLRs of redundancy version rv to the kept buffers and leaves blocks whose
// CRC already passed alone. UCI (ACK / RI / CQI) is per transmission and decoded afresh on every call.
extern "C" int srslte_hip_ul_rx_batch_harq(srslte_hip_ul_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, uint32_t rv, int new_data,
                                           uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream)
{
  if (!q || !d_iq || !d_tb || !d_tb_ok || rv > 3 || nof_sf > q->cfg.max_batch || tb_stride < q->cfg.tbs / 8 + 6) return SRSLTE_ERROR_INVALID_INPUTS;
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  hipStream_t    st = (hipStream_t)stream;
  const uint32_t C = q->seg.C, K = q->seg.K1;
  const uint32_t* d_rm_tbl = q->d_rm_tbl;
  if (rv) {
    if (!q->d_rm_tbl_rv[rv]) {
      std::vector<uint32_t> t;
      lte_rm_rx_table(K, rv, t);
      if (q->W) {
        for (auto& v : t) {
          v = v < 3 * K ? (v % 3) * (K + 32) + ((v / 3) % (K / q->W)) * q->W + (v / 3) / (K / q->W) : (v - 3 * K) + 3 * (K + 32);
        }
      }
      if (int rc = upload(&q->d_rm_tbl_rv[rv], rm_slot_table(t, q->in_stride))) return rc;
    }
    d_rm_tbl = q->d_rm_tbl_rv[rv];
  }
  const int combine = new_data ? 0 : 1;
  int            r = srslte_hip_ofdm_rx_sf_batch(q->ofdm, d_iq, q->d_grid, (int)nof_sf, stream);
  if (r) return r;
  r = srslte_hip_chest_ul_estimate_pusch_batch_hop(q->chest, tti0, q->cfg.L_prb, q->cfg.n_prb, (uint32_t)q->pg.n_prb1, q->cfg.n_dmrs, q->d_grid, q->d_ce, q->d_res, (int)nof_sf,
                                               stream);
  if (r) return r;
  PuschGeom g = q->pg;
  g.tti0      = (int)tti0;
  g.ack_sum   = q->d_ack_sum;
  g.ri_sum    = q->d_ack_sum + 4 * q->cfg.max_batch;
  const dim3 grid(ceil_div(g.M_sc, 256), g.nsymb, nof_sf);
  hipLaunchKernelGGL(pusch_eq_kernel, grid, dim3(256), 0, st, (const cf32*)q->d_grid, (const cf32*)q->d_ce, (const float*)q->d_res, q->d_z, g);
  LAUNCH_CHECK();
  r = srslte_hip_dft_precoding_batch(q->d_z, q->d_d, q->cfg.L_prb, g.nsymb * nof_sf, 0, stream); // srslte_dft_precoding_init_rx: inverse, 1/sqrt(N)
  if (r) return r;
  if (g.ack.O) HIP_TRY(hipMemsetAsync(q->d_ack_sum, 0, sizeof(int) * 4 * nof_sf, st));
  if (g.ri.O) HIP_TRY(hipMemsetAsync(q->d_ack_sum + 4 * q->cfg.max_batch, 0, sizeof(int) * 4 * nof_sf, st));
  hipLaunchKernelGGL(pusch_demod_kernel, dim3(ceil_div(g.M_sc, 64), nof_sf), dim3(256), 0, st, (const cf32*)q->d_d, (const uint32_t*)q->d_scr, q->d_g, g);
  LAUNCH_CHECK();
  if (g.ack.O) {
    hipLaunchKernelGGL(pusch_ack_decide_kernel, dim3(ceil_div((int)nof_sf, 64)), dim3(64), 0, st, (const int*)q->d_ack_sum, q->d_ack, (int)nof_sf);
    LAUNCH_CHECK();
  }
  if (g.ri.O) {
    hipLaunchKernelGGL(pusch_ack_decide_kernel, dim3(ceil_div((int)nof_sf, 64)), dim3(64), 0, st, (const int*)(q->d_ack_sum + 4 * q->cfg.max_batch),
                       q->d_ack + 2 * q->cfg.max_batch, (int)nof_sf);
    LAUNCH_CHECK();
  }
  if (q->cfg.cqi_len) { // the report in front of the UL-SCH (sch.c:1031-1056)
    hipLaunchKernelGGL(pusch_cqi_decode_kernel, dim3(nof_sf), dim3(256), 0, st, (const int16_t*)q->d_g, q->rg.max_bits, q->Qp_cqi * q->rg.Qm,
                       (int)q->cfg.cqi_len, q->d_cqi, q->d_cqi + 64 * q->cfg.max_batch, (const PuschDesc*)nullptr);
    LAUNCH_CHECK();
  }
  RmGeom rg = q->rg;
  rg.tti0   = (int)tti0;
  rg.combine = combine;
  rg.skip    = combine ? q->d_cb_ok : nullptr;
  if (rm_fits_lds(rg)) {
    hipLaunchKernelGGL(rm_rx_lds_kernel<int16_t>, dim3(nof_sf * C), dim3(256), rm_lds_bytes(rg, 2), st, (const int16_t*)q->d_g, q->d_w, d_rm_tbl, rg);
  } else {
    hipLaunchKernelGGL(rm_rx_kernel<int16_t>, dim3(ceil_div(rg.w_stride, 512), nof_sf * C), dim3(256), 0, st, (const int16_t*)q->d_g, q->d_w,
                       d_rm_tbl, rg);
  }
  LAUNCH_CHECK();
  tdec_set_tb_syndrome(q->tdec, q->d_tb_rem, C, q->d_cb_syn);
  tdec_set_skip(q->tdec, combine ? q->d_cb_ok : nullptr);
  r = tdec_run_batch_w(q->tdec, q->d_w, 0, q->in_stride, q->W != 0, K, -1, nof_sf * C, q->cfg.max_iterations, C > 1 ? 0x1800063u : 0x1864CFBu,
                       C > 1 ? K : q->cfg.tbs + 24, q->d_cb_bytes, K / 8, q->d_cb_iters, q->d_cb_ok, st);
  if (r) return r;
  TbGeom tg    = q->tg;
  tg.tb_stride = (int)tb_stride;
  if (q->d_tb_rem) {
    hipLaunchKernelGGL(tb_asm_kernel, dim3(nof_sf), dim3(256), 0, st, (const uint8_t*)q->d_cb_bytes, (const uint8_t*)q->d_cb_ok,
                       (const uint32_t*)q->d_cb_syn, d_tb, d_tb_ok, tg);
  } else {
    hipLaunchKernelGGL(tb_crc_kernel, dim3(nof_sf), dim3(512), 0, st, (const uint8_t*)q->d_cb_bytes, (const uint8_t*)q->d_cb_ok,
                       (const uint32_t*)q->d_tbcrc, d_tb, d_tb_ok, tg);
  }
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}

static int ul_rx_grants_init(srslte_hip_ul_rx_t* q, uint32_t V, uint32_t max_re)
{
  const uint32_t P = q->cfg.nof_prb;
  q->gs = new GrantsState();
  if (grants_alloc(q->gs, 12 * 12 * P, V, q->seg.C, 0, false, (sizeof(PuschDesc) + sizeof(ChestUlItem)) * V)) return SRSLTE_ERROR;
  HIP_TRY(hipMalloc((void**)&q->g_z, sizeof(cf32) * (size_t)max_re * V));
  HIP_TRY(hipMalloc((void**)&q->g_d, sizeof(cf32) * (size_t)max_re * V));
  HIP_TRY(hipMalloc((void**)&q->g_res, sizeof(float) * 5 * V));
  HIP_TRY(hipMalloc((void**)&q->g_uci_sum, sizeof(int) * 8 * V));
  HIP_TRY(hipMalloc((void**)&q->g_uci, (size_t)4 * V));
  HIP_TRY(hipMalloc((void**)&q->g_cqi, (size_t)65 * V));
  HIP_TRY(hipMemset(q->g_cqi, 0, (size_t)65 * V));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(q->g_uci, 0, (size_t)4 * V));
  HIP_TRY(hipDeviceSynchronize());
  return SRSLTE_SUCCESS;
}

// Per-PUSCH grants: what an eNB receives in a run of TTIs - any number of PUSCHs per subframe, each with its own allocation (L_prb, PRB offset per
// slot), DMRS cyclic shift, RNTI, modulation, transport block and redundancy version (srslte_enb_ul_get_pusch called once per scheduled UE,
// enb_ul.c:200-235, after one srslte_enb_ul_fft per TTI). The OFDM demodulation runs once per subframe; estimator, equaliser and demapper take
// their geometry from per-PUSCH descriptors, the transform de-precoding runs once per distinct L_prb (PUSCHs of one size sit next to each other
// in the symbol buffers), and from the LLRs on it is the downlink's grants machinery with one slot per PUSCH: slot p = grants[p] keeps soft
// buffers, CRC flags and bytes between calls (HARQ as srslte_hip_ul_rx_batch_harq). HARQ-ACK, rank indication and CQI report per PUSCH.
extern "C" int srslte_hip_ul_rx_batch_grants(srslte_hip_ul_rx_t* q, const void* d_iq, uint32_t tti0, uint32_t nof_sf, const srslte_hip_ul_grant_t* grants,
                                             uint32_t nof_grants, uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, void* stream)
{
  if (!q || !d_iq || !grants || !d_tb || !d_tb_ok || nof_sf > q->cfg.max_batch || tb_stride < q->cfg.tbs / 8 + 6) return SRSLTE_ERROR_INVALID_INPUTS;
  const uint32_t V = q->cfg.max_grants ? q->cfg.max_grants : q->cfg.max_batch, P = q->cfg.nof_prb;
  if (nof_grants > V) return SRSLTE_ERROR_INVALID_INPUTS;
  if (nof_sf == 0 || nof_grants == 0) return SRSLTE_SUCCESS;
  hipStream_t    st    = (hipStream_t)stream;
  const uint32_t nsymb = (uint32_t)q->pg.nsymb, max_re = nsymb * 12 * P;
  if (!q->gs && ul_rx_grants_init(q, V, max_re)) { // a failed start leaves no half-made state behind
    grants_free(q->gs);
    q->gs = nullptr;
    return SRSLTE_ERROR;
  }
  GrantsState*   g    = q->gs;
  const size_t   nblk = (size_t)V * g->Cmax;
  const uint32_t hs   = g->h_slot++ & 3u;
  if (g->h_used[hs]) HIP_TRY(hipEventSynchronize(g->h_ev[hs])); // the copy that last read this buffer (four calls ago) has completed
  auto* h_gr = reinterpret_cast<GrantDev*>(g->h_pin[hs]);
  auto* h_sf = reinterpret_cast<SfDesc*>(h_gr + V);
  auto* h_cb = reinterpret_cast<CbDesc*>(h_sf + V);
  auto* h_map = reinterpret_cast<uint32_t*>(h_cb + nblk);
  auto* h_pd = reinterpret_cast<PuschDesc*>(h_map + nblk);
  auto* h_it = reinterpret_cast<ChestUlItem*>(h_pd + V);
  auto* d_gr = reinterpret_cast<GrantDev*>(g->d_desc);
  auto* d_sf = reinterpret_cast<SfDesc*>(d_gr + V);
  auto* d_cb = reinterpret_cast<CbDesc*>(d_sf + V);
  auto* d_map = reinterpret_cast<uint32_t*>(d_cb + nblk);
  auto* d_pd = reinterpret_cast<PuschDesc*>(d_map + nblk);
  auto* d_it = reinterpret_cast<ChestUlItem*>(d_pd + V);
  GrantsBuild bd;
  bd.g = g; bd.h_sf = h_sf; bd.h_cb = h_cb; bd.l8 = false; bd.max_tbs = q->cfg.tbs; bd.npt = 1; bd.max_mod = 3; bd.who = "ul_rx";
  // PUSCHs in the order (L_prb, n_dmrs): one estimator launch per (L_prb, n_dmrs), one de-precoding launch per L_prb
  std::vector<uint32_t> order(nof_grants);
  for (uint32_t p = 0; p < nof_grants; p++) order[p] = p;
  uint32_t max_M = 0;
  for (uint32_t p = 0; p < nof_grants; p++) {
    const srslte_hip_ul_grant_t& gr = grants[p];
    if (gr.sf >= nof_sf || gr.L_prb == 0 || !srslte_hip_dft_precoding_valid_prb(gr.L_prb) || gr.n_prb + gr.L_prb > P || gr.n_prb_slot1 + gr.L_prb > P ||
        gr.n_dmrs >= 8) {
      hip_log("[srslte_hip] ul_rx grants: entry %u: invalid allocation (subframe %u of %u, L_prb %u at %u / %u of %u PRB, n_dmrs %u)\n", p, gr.sf, nof_sf, gr.L_prb,
              gr.n_prb, gr.n_prb_slot1, P, gr.n_dmrs);
      return SRSLTE_ERROR_INVALID_INPUTS;
    }
    max_M = 12 * gr.L_prb > max_M ? 12 * gr.L_prb : max_M;
  }
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    return grants[a].L_prb != grants[b].L_prb ? grants[a].L_prb < grants[b].L_prb : grants[a].n_dmrs < grants[b].n_dmrs;
  });
  uint32_t zoff = 0;
  bool     any_uci = false, any_cqi = false;
  for (uint32_t i = 0; i < nof_grants; i++) {
    const uint32_t               p  = order[i];
    const srslte_hip_ul_grant_t& gr = grants[p];
    GrantDev&                    gd = h_gr[p];
    memset(&gd, 0, sizeof(gd));
    memset(&h_sf[p], 0, sizeof(SfDesc));
    gd.sf_idx = (int)((tti0 + gr.sf) % 10); gd.rnti = gr.rnti; // srslte_sequence_pusch (sequences.c:65-67): the PDSCH's c_init with q = 0
    h_sf[p].scr = g->d_scr + (size_t)p * g->words;
    PuschDesc& pd = h_pd[p];
    memset(&pd, 0, sizeof(pd));
    pd.sf = (int)gr.sf; pd.M_sc = 12 * (int)gr.L_prb; pd.n_prb = (int)gr.n_prb; pd.n_prb1 = (int)gr.n_prb_slot1; pd.mod = gr.mod; pd.Qm = 2 * gr.mod;
    pd.zoff = (int)zoff;
    zoff += nsymb * 12 * gr.L_prb;
    h_it[i].sf = (int)gr.sf; h_it[i].n_prb = (int)gr.n_prb; h_it[i].n_prb1 = (int)gr.n_prb_slot1; h_it[i].row = (int)p;
    // HARQ-ACK and rank indication of this PUSCH (srslte_uci_cfg_t of its srslte_pusch_cfg_t): Q' from the grant's own size and code blocks; the
    // UL-SCH is rate-matched to what the RI symbols leave (sch.c:1157-1160), the ACK symbols overwrite it
    srslte_hip_cbsegm_t seg;
    const uint32_t      nof_re = nsymb * 12 * gr.L_prb;
    if (srslte_hip_cbsegm(&seg, gr.tbs)) return SRSLTE_ERROR_INVALID_INPUTS;
    const int Qp_ack = pusch_ack_qprime(gr.ack_len, gr.I_offset_ack, gr.L_prb, nsymb, seg.C * seg.K1);
    const int Qp_ri  = pusch_ack_qprime(gr.ri_len, gr.I_offset_ri, gr.L_prb, nsymb, seg.C * seg.K1, true);
    const int Qp_cqi = Qp_ri >= 0 && gr.cqi_len <= 64 ? pusch_cqi_qprime(gr.cqi_len, gr.I_offset_cqi, gr.L_prb, nsymb, seg.C * seg.K1, (uint32_t)Qp_ri) : -1;
    if (Qp_ack < 0 || Qp_ri < 0 || Qp_cqi < 0 || (uint32_t)(Qp_ri + Qp_cqi) + seg.C >= nof_re || gr.mod < 1 || gr.mod > 3) {
      hip_log("[srslte_hip] ul_rx grants: entry %u: invalid UCI configuration (ack %u / %u, ri %u / %u, cqi %u / %u)\n", p, gr.ack_len, gr.I_offset_ack, gr.ri_len,
              gr.I_offset_ri, gr.cqi_len, gr.I_offset_cqi);
      return SRSLTE_ERROR_INVALID_INPUTS;
    }
    pd.ack.O = (int)gr.ack_len; pd.ack.Qprime = Qp_ack; pd.ri.O = (int)gr.ri_len; pd.ri.Qprime = Qp_ri;
    pd.cqi_O = (int)gr.cqi_len; pd.cqi_Q = Qp_cqi * 2 * gr.mod;
    any_uci = any_uci || gr.ack_len || gr.ri_len;
    any_cqi = any_cqi || gr.cqi_len;
    // the report's LLRs come first in the row; the UL-SCH is rate-matched to the rest (sch.c:1058-1064)
    if (int r = bd.add_tb(p, p, gr.mod, gr.tbs, gr.rv, gr.new_data, nof_re - (uint32_t)Qp_ri - (uint32_t)Qp_cqi, 1, (uint32_t)pd.cqi_Q)) return r;
  }
  bd.fill_map(h_map);
  int r = srslte_hip_ofdm_rx_sf_batch(q->ofdm, d_iq, q->d_grid, (int)nof_sf, stream);
  if (r) return r;
  HIP_TRY(hipMemcpyAsync(g->d_desc, g->h_pin[hs], g->desc_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(g->h_ev[hs], st));
  g->h_used[hs] = true;
  for (uint32_t i = 0; i < nof_grants;) { // estimator: runs of equal (L_prb, n_dmrs)
    uint32_t j = i + 1;
    while (j < nof_grants && grants[order[j]].L_prb == grants[order[i]].L_prb && grants[order[j]].n_dmrs == grants[order[i]].n_dmrs) j++;
    r = chest_ul_estimate_items(q->chest, tti0, grants[order[i]].L_prb, grants[order[i]].n_dmrs, d_it + i, (int)(j - i), q->d_grid, q->d_ce, q->g_res, st);
    if (r) return r;
    i = j;
  }
  hipLaunchKernelGGL(scr_gen_kernel, dim3(ceil_div((int)g->words, 256), nof_grants), dim3(256), 0, st, (const GrantDev*)d_gr, (const uint32_t*)g->d_basis, g->d_scr,
                     (int)g->words, (int)q->cfg.cell_id);
  hipLaunchKernelGGL(pusch_eq_grants_kernel, dim3(ceil_div((int)max_M, 256), nsymb, nof_grants), dim3(256), 0, st, (const cf32*)q->d_grid, (const cf32*)q->d_ce,
                     (const float*)q->g_res, q->g_z, (const PuschDesc*)d_pd, 12 * (int)P, (int)nsymb, q->cfg.mmse);
  LAUNCH_CHECK();
  for (uint32_t i = 0; i < nof_grants;) { // inverse transform precoding: runs of equal L_prb (srslte_dft_precoding_init_rx: inverse, 1/sqrt(N))
    uint32_t j = i + 1;
    while (j < nof_grants && grants[order[j]].L_prb == grants[order[i]].L_prb) j++;
    const size_t off = (size_t)h_pd[order[i]].zoff;
    r = srslte_hip_dft_precoding_batch(q->g_z + off, q->g_d + off, grants[order[i]].L_prb, nsymb * (j - i), 0, stream);
    if (r) return r;
    i = j;
  }
  if (any_uci) HIP_TRY(hipMemsetAsync(q->g_uci_sum, 0, sizeof(int) * 8 * V, st));
  hipLaunchKernelGGL(pusch_demod_grants_kernel, dim3(ceil_div((int)max_M, 64), nof_grants), dim3(256), 0, st, (const cf32*)q->g_d, (const uint32_t*)g->d_scr,
                     (int)g->words, g->d_e, (int)g->max_bits, (const PuschDesc*)d_pd, (int)nsymb, q->g_uci_sum, q->g_uci_sum + 4 * V);
  LAUNCH_CHECK();
  if (any_uci) { // decisions of every row of the call (zero sums -> 0 where a PUSCH carries none); a call without any leaves the rows alone
    hipLaunchKernelGGL(pusch_ack_decide_kernel, dim3(ceil_div((int)nof_grants, 64)), dim3(64), 0, st, (const int*)q->g_uci_sum, q->g_uci, (int)nof_grants);
    hipLaunchKernelGGL(pusch_ack_decide_kernel, dim3(ceil_div((int)nof_grants, 64)), dim3(64), 0, st, (const int*)(q->g_uci_sum + 4 * V), q->g_uci + 2 * V,
                       (int)nof_grants);
  }
  if (any_cqi) { // the reports in front of the UL-SCH (sch.c:1031-1056); rows without one keep what they held
    hipLaunchKernelGGL(pusch_cqi_decode_kernel, dim3(nof_grants), dim3(256), 0, st, (const int16_t*)g->d_e, (int)g->max_bits, 0, 0, q->g_cqi, q->g_cqi + 64 * V,
                       (const PuschDesc*)d_pd);
  }
  LAUNCH_CHECK();
  return grants_back_end(g, bd, d_sf, d_cb, d_map, tti0, q->cfg.max_iterations, nof_grants, nof_grants, 0, d_tb, tb_stride, d_tb_ok, st);
}

// ====================================================================================================================
// PUSCH transmit pipeline (UE side): TB CRC24A + segmentation + CB CRC24B (sch.c:183-297 as used by
// srslte_ulsch_encode :1068-1160) -> turbo encoder -> rate matching + UL channel interleaver + scrambling + modulation
// (rm_turbo.c:100-158, sch.c:580-598, pusch.c:380-400) -> transform precoding (:406) -> RE mapping with the DMRS
// (pusch_put :52-91, ue_ul.c:320-326) -> OFDM TX with 1/sqrt(N) and the +1/2 carrier shift (ue_ul.c:59-64).
// Same restrictions as the receive side: UL-SCH data only, same allocation in both slots, normal CP, rv 0.
// ====================================================================================================================
namespace {


struct PuschTxGeom {
  int   cell_nre, M_sc, n_prb, n_prb1, Qm, tti0, scr_words, C, K, tbs, rlenB, cb_stride, par_stride, tb_stride, rm_len, syms_lo, C_lo;
  int   nsymb; // 12 data symbols, 11 in a shortened subframe
  AckGeom        ack, ri;
  const uint8_t* ack_bits; // [nof_sf][2] HARQ-ACK values of this call, or null
  const uint8_t* ri_bits;  // [nof_sf][2] rank indication bits of this call, or null
  const uint8_t* q_cqi;    // [nof_sf][cqi_stride] coded CQI report bits of this call, or null
  int            Qp_cqi, cqi_stride;
  float lvl[16];
};

// grid = nof_sf, 256 threads: the coded CQI / PMI report, Q = Q' Qm bits per subframe (srslte_uci_encode_cqi_pusch, uci.c:470-494). Up to 11
// bits: the (32, O) block code repeated (encode_cqi_short :283-302). Above: CRC-8, the tail-biting rate-1/3 convolutional code (a circular
// convolution with 0x6D, 0x4F, 0x57; convcoder.c:43-72) and srslte_rm_conv_tx (rm_conv.c:44-89) as the table rm: output bit -> coded bit.
__global__ __launch_bounds__(256) void pusch_cqi_encode_kernel(const uint8_t* __restrict__ cqi, uint8_t* __restrict__ qb, int q_stride, int Q, int O,
                                                               const uint16_t* __restrict__ rm)
{
  __shared__ uint8_t msg[72], enc[3 * 72];
  const int          sf = blockIdx.x, tid = threadIdx.x;
  const uint8_t*     in = cqi + (size_t)sf * 64;
  uint8_t*           out = qb + (size_t)sf * q_stride;
  if (O <= 11) {
    if (tid < 32) {
      int b = 0;
      for (int n = 0; n < O; n++) b ^= in[n] & (CQI_M32[tid] >> n) & 1;
      enc[tid] = (uint8_t)b;
    }
    __syncthreads();
    for (int i = tid; i < Q; i += 256) out[i] = enc[i & 31];
    return;
  }
  const int F = O + 8;
  if (tid < O) msg[tid] = in[tid] & 1;
  __syncthreads();
  if (tid == 0) {
    const uint32_t c = cqi_crc8(msg, O);
    for (int i = 0; i < 8; i++) msg[O + i] = (c >> (7 - i)) & 1;
  }
  __syncthreads();
  for (int e = tid; e < 3 * F; e += 256) {
    const int      i = e / 3, p = e - 3 * i;
    const uint32_t poly = p == 0 ? 0x6Du : (p == 1 ? 0x4Fu : 0x57u);
    int            b = 0;
    for (int j = 0; j < 7; j++) b ^= ((poly >> j) & 1u) & msg[(i - j + F) % F];
    enc[e] = (uint8_t)b;
  }
  __syncthreads();
  for (int i = tid; i < Q; i += 256) out[i] = enc[rm[i]];
}

// grid = nof_sf, 256 threads: CRC24A of each transport block (sch.c:470-488 on the transmit side :1104-1110)
__global__ __launch_bounds__(256) void pusch_tx_tbcrc_kernel(const uint8_t* __restrict__ tb, uint32_t* __restrict__ crc_out, PuschTxGeom g)
{
  __shared__ uint32_t tab[256], red[256];
  const uint8_t*      x = tb + (size_t)blockIdx.x * g.tb_stride;
  const uint32_t      c = block_crc24([&](int i) { return (uint32_t)x[i]; }, g.tbs / 8, 0x1864CFBu, tab, red);
  if (threadIdx.x == 0) crc_out[blockIdx.x] = c;
}

// grid = (C, nof_sf), 256 threads: code block r = bytes [r*rlenB, (r+1)*rlenB) of TB | CRC24A, then its CRC24B when C > 1 (sch.c:222-262)
__global__ __launch_bounds__(256) void pusch_tx_seg_kernel(const uint8_t* __restrict__ tb, const uint32_t* __restrict__ tbcrc, uint8_t* __restrict__ cb,
                                                           PuschTxGeom g)
{
  __shared__ uint32_t tab[256], red[256];
  __shared__ uint8_t  xs[768];
  const int           r = blockIdx.x, sf = blockIdx.y, t = threadIdx.x, tbB = g.tbs / 8;
  const uint8_t*      x   = tb + (size_t)sf * g.tb_stride;
  const uint32_t      crc = tbcrc[sf];
  uint8_t*            out = cb + ((size_t)sf * g.C + r) * g.cb_stride;
  for (int i = t; i < g.rlenB; i += 256) {
    const int j = r * g.rlenB + i;
    xs[i]       = j < tbB ? x[j] : (uint8_t)(crc >> (8 * (2 - (j - tbB))));
  }
  __syncthreads();
  for (int i = t; i < g.rlenB; i += 256) out[i] = xs[i];
  if (g.C > 1) {
    const uint32_t c = block_crc24([&](int i) { return (uint32_t)xs[i]; }, g.rlenB, 0x1800063u, tab, red);
    if (t < 3) out[g.rlenB + t] = (uint8_t)(c >> (8 * (2 - t)));
  }
}

// grid = (ceil(M_sc/256), 12, nof_sf): modulation symbol (n, k) of the interleaved, scrambled stream: its Qm bits are
// g[(k*12 + n)*Qm + b] (36.212 5.2.2.8 without UCI), all from one code block; bit e of a block = coded bit rm[e mod (3K+12)]
// (rm: circular-buffer order with the NULLs removed; source 0 = systematic byte stream, 1 = its tail nibble, 2 = parity stream)
// One PUSCH: cs its scrambling bits, cb0 its first code-block slot, ack / ri / qc its UCI inputs (or null), d its [nsymb][M_sc] symbols, lvl the
// levels of its modulation (the geometry's own lvl member is not read here)
__device__ __forceinline__ void pusch_tx_mod_body(const uint8_t* __restrict__ cb, const uint8_t* __restrict__ parity, const uint8_t* __restrict__ sys_tail,
                                                  const uint32_t* __restrict__ rm, const uint32_t* __restrict__ cs, cf32* __restrict__ d, const PuschTxGeom& g,
                                                  int k, int n, int cb0, const uint8_t* __restrict__ ack, const uint8_t* __restrict__ rib,
                                                  const uint8_t* __restrict__ qc, const float* __restrict__ lvl)
{
  if (k >= g.M_sc) return;
  const int ri = rib ? ri_symbol_index(g.ri, n, k, g.M_sc, g.nsymb) : -1; // RI symbol: outside the UL-SCH stream (sch.c:580-598)
  // symbol index in g order; blocks 0..C_lo-1 carry syms_lo symbols, the rest syms_lo + 1 (sch.c:238-243)
  const int s0 = ri >= 0 ? 0 : k * g.nsymb + n - (rib ? ri_before(g.ri, n, k, g.M_sc, g.nsymb) : 0);
  // the first Q'_cqi symbols of the stream carry the CQI report, the UL-SCH follows (sch.c:1133-1160)
  const bool is_cqi = ri < 0 && s0 < g.Qp_cqi;
  const int  s      = is_cqi || ri >= 0 ? 0 : s0 - g.Qp_cqi;
  int        r, e0;
  if (s < g.C_lo * g.syms_lo) {
    r  = s / g.syms_lo;
    e0 = (s - r * g.syms_lo) * g.Qm;
  } else {
    const int u = s - g.C_lo * g.syms_lo;
    r           = g.C_lo + u / (g.syms_lo + 1);
    e0          = (u % (g.syms_lo + 1)) * g.Qm;
  }
  const size_t    cbi = (size_t)cb0 + r;
  const uint8_t * xb = cb + cbi * g.cb_stride, *pb = parity + cbi * g.par_stride;
  const int       q0  = (n * g.M_sc + k) * g.Qm;
  const int       ai  = ack ? ack_symbol_index(g.ack, n, k, g.M_sc, g.nsymb) : -1;
  int             re = 0, im = 0, prev = 0;
  for (int b = 0; b < g.Qm; b++) {
    const uint32_t src = rm[(e0 + b) % g.rm_len], pos = src & 0x3fffffffu;
    const uint8_t  byte = (src >> 30) == 0 ? xb[pos >> 3] : ((src >> 30) == 1 ? sys_tail[cbi] : pb[pos >> 3]);
    int            bit  = (byte >> (7 - (pos & 7))) & 1;
    const int      cbit = (cs[(q0 + b) >> 5] >> ((q0 + b) & 31)) & 1;
    if (is_cqi) bit = qc[s0 * g.Qm + b];
    bit ^= cbit;
    if (ai >= 0) { // HARQ-ACK symbol: value bits are scrambled, placeholders are 1, a repetition bit copies the transmitted bit before it
      const int t = ack_bit_type(ack, g.ack.O, g.Qm, ai * g.Qm + b); // (sch.c:1203-1215, pusch.c:386-400)
      bit         = t == 3 ? 1 : (t == 2 ? prev : (t ^ cbit));
    }
    if (ri >= 0) { // rank indication: the same encoder (sch.c:1110-1129)
      const int t = ack_bit_type(rib, g.ri.O, g.Qm, ri * g.Qm + b);
      bit         = t == 3 ? 1 : (t == 2 ? prev : (t ^ cbit));
    }
    prev = bit;
    if (b & 1) im = (im << 1) | bit;
    else re = (re << 1) | bit;
  }
  d[(size_t)n * g.M_sc + k] = make_float2(lvl[re], lvl[im]);
}


__global__ __launch_bounds__(256) void pusch_tx_mod_kernel(const uint8_t* __restrict__ cb, const uint8_t* __restrict__ parity,
                                                           const uint8_t* __restrict__ sys_tail, const uint32_t* __restrict__ rm,
                                                           const uint32_t* __restrict__ scr, cf32* __restrict__ d, PuschTxGeom g)
{
  const int sf = blockIdx.z, sf_idx = (g.tti0 + sf) % 10;
  pusch_tx_mod_body(cb, parity, sys_tail, rm, scr + (size_t)sf_idx * g.scr_words, d + (size_t)sf * g.nsymb * g.M_sc, g, blockIdx.x * blockDim.x + threadIdx.x,
                    blockIdx.y, sf * g.C, g.ack_bits ? g.ack_bits + 2 * sf : nullptr, g.ri_bits ? g.ri_bits + 2 * sf : nullptr,
                    g.q_cqi ? g.q_cqi + (size_t)sf * g.cqi_stride : nullptr, g.lvl);
}

// grid = (ceil(cell_nre/256), 14, nof_sf): resource grid of the subframe: z on the granted PRBs of the 12 data symbols, DMRS on
// l = 3, 10, zero elsewhere (the caller's memset + pusch_put + srslte_refsignal_dmrs_pusch_put)
__global__ __launch_bounds__(256) void pusch_tx_map_kernel(const cf32* __restrict__ z, const cf32* __restrict__ dmrs, cf32* __restrict__ grid,
                                                           PuschTxGeom g)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y, sf = blockIdx.z, sf_idx = (g.tti0 + sf) % 10;
  if (k >= g.cell_nre) return;
  const int kk = k - 12 * (l < 7 ? g.n_prb : g.n_prb1); // each slot at its own offset (pusch_cp, pusch.c:52-91; refsignal_ul.c:316-330)
  cf32      v  = make_float2(0.f, 0.f);
  if (kk >= 0 && kk < g.M_sc) {
    if (l == 3 || l == 10) {
      v = dmrs[((size_t)sf_idx * 2 + (l == 10)) * g.M_sc + kk];
    } else {
      const int n = l < 3 ? l : (l < 10 ? l - 1 : l - 2);
      if (n < g.nsymb) v = z[((size_t)sf * g.nsymb + n) * g.M_sc + kk]; // the last symbol of a shortened subframe stays empty
    }
  }
  grid[((size_t)sf * 14 + l) * g.cell_nre + k] = v;
}

} // namespace

struct srslte_hip_ul_tx {
  srslte_hip_ul_tx_cfg_t cfg;
  srslte_hip_ofdm_t*     ofdm;
  srslte_hip_chest_ul_t* dmrs;
  srslte_hip_cbsegm_t    seg;
  PuschTxGeom            g;
  uint32_t *             d_scr, *d_rm, *d_tbcrc;
  uint32_t*              d_rm_rv[4]; // rate-matching tables of redundancy versions 1-3, made on first use ([0] unused: d_rm)
  uint8_t *              d_cb, *d_parity, *d_sys_tail, *d_qcqi;
  uint16_t*              d_cqi_rm;
  cf32 *                 d_d, *d_z, *d_grid;
  struct UlTxGrantsState* gs; // srslte_hip_ul_tx_batch_grants: created on first use
};
static void ul_tx_grants_free(struct UlTxGrantsState* g);

extern "C" void srslte_hip_ul_tx_destroy(srslte_hip_ul_tx_t* q)
{
  if (!q) return;
  srslte_hip_ofdm_destroy(q->ofdm);
  srslte_hip_chest_ul_destroy(q->dmrs);
  void* bufs[] = {q->d_scr, q->d_rm, q->d_tbcrc, q->d_cb, q->d_parity, q->d_sys_tail, q->d_d, q->d_z, q->d_grid, q->d_qcqi, q->d_cqi_rm,
                  q->d_rm_rv[1], q->d_rm_rv[2], q->d_rm_rv[3]};
  for (void* b : bufs) {
    if (b) (void)hipFree(b);
  }
  ul_tx_grants_free(q->gs);
  delete q;
}

extern "C" srslte_hip_ul_tx_t* srslte_hip_ul_tx_create(const srslte_hip_ul_tx_cfg_t* cfg)
{
  if (!cfg || cfg->max_batch == 0 || cfg->mod < 1 || cfg->mod > 3 || cfg->L_prb < 1 || cfg->n_prb + cfg->L_prb > cfg->nof_prb ||
      (cfg->hopping && cfg->n_prb_slot1 + cfg->L_prb > cfg->nof_prb) ||
      !srslte_hip_dft_precoding_valid_prb(cfg->L_prb)) {
    hip_log("[srslte_hip] ul_tx: invalid configuration\n");
    return nullptr;
  }
  auto* q = new srslte_hip_ul_tx();
  memset(q, 0, sizeof(*q));
  q->cfg = *cfg;
  if (srslte_hip_cbsegm(&q->seg, cfg->tbs) || q->seg.F || q->seg.C2 || (cfg->tbs % 8)) {
    hip_log("[srslte_hip] ul_tx: TBS %u needs filler bits or two code-block sizes; not supported on device yet\n", cfg->tbs);
    delete q;
    return nullptr;
  }
  const uint32_t P = cfg->nof_prb, B = cfg->max_batch, C = q->seg.C, K = q->seg.K1, Qm = 2 * (uint32_t)cfg->mod, M_sc = 12 * cfg->L_prb;
  const uint32_t nsymb = cfg->shortened ? 11 : 12;
  const uint32_t nof_re = nsymb * M_sc, nbits = nof_re * Qm, scr_words = (nbits + 31) / 32;
  PuschTxGeom&   g = q->g;
  g.cell_nre = 12 * (int)P; g.M_sc = (int)M_sc; g.n_prb = (int)cfg->n_prb; g.n_prb1 = (int)(cfg->hopping ? cfg->n_prb_slot1 : cfg->n_prb); g.Qm = (int)Qm; g.scr_words = (int)scr_words; g.C = (int)C; g.K = (int)K;
  g.nsymb = (int)nsymb;
  g.ack.O = (int)cfg->ack_len; g.ack.Qprime = pusch_ack_qprime(cfg->ack_len, cfg->I_offset_ack, cfg->L_prb, nsymb, C * K);
  if (g.ack.Qprime < 0) {
    hip_log("[srslte_hip] ul_tx: invalid HARQ-ACK configuration\n");
    delete q;
    return nullptr;
  }
  g.tbs = (int)cfg->tbs; g.rlenB = (int)((C == 1 ? K : K - 24) / 8); g.cb_stride = (int)((K / 8 + 15) & ~15u);
  g.par_stride = (int)((K / 4 + 1 + 15) & ~15u); g.rm_len = (int)(3 * K + 12);
  g.ri.O = (int)cfg->ri_len; g.ri.Qprime = pusch_ack_qprime(cfg->ri_len, cfg->I_offset_ri, cfg->L_prb, nsymb, C * K, true);
  if (g.ri.Qprime < 0 || (uint32_t)g.ri.Qprime >= nof_re) {
    hip_log("[srslte_hip] ul_tx: invalid rank-indication configuration\n");
    delete q;
    return nullptr;
  }
  g.Qp_cqi = pusch_cqi_qprime(cfg->cqi_len, cfg->I_offset_cqi, cfg->L_prb, nsymb, C * K, (uint32_t)g.ri.Qprime);
  if (g.Qp_cqi < 0 || (uint32_t)(g.ri.Qprime + g.Qp_cqi) + C >= nof_re) {
    hip_log("[srslte_hip] ul_tx: invalid CQI configuration\n");
    delete q;
    return nullptr;
  }
  g.cqi_stride = (g.Qp_cqi * (int)Qm + 15) & ~15;
  const uint32_t g_re = nof_re - g.ri.Qprime - g.Qp_cqi; // UL-SCH symbols: what the RI and the CQI report leave (sch.c:1157-1160)
  g.syms_lo = (int)(g_re / C); g.C_lo = (int)(C - g_re % C); // G' = the UL-SCH symbols, gamma = G' mod C (sch.c:205-207)
  for (uint32_t idx = 0; idx < (1u << cfg->mod); idx++) { // 36.211 7.1.2-7.1.4, one axis: bits b0 b2 b4 of the symbol (lte_tables.c:57-182)
    const int    nb = cfg->mod;
    double       v  = 1.0;
    for (int i = nb - 1; i >= 1; i--) v = (double)(1 << (nb - i)) - (1 - 2 * (int)((idx >> (nb - 1 - i)) & 1)) * v;
    const double norm = nb == 1 ? sqrt(2.0) : (nb == 2 ? sqrt(10.0) : sqrt(42.0));
    g.lvl[idx]        = (float)((1 - 2 * (int)((idx >> (nb - 1)) & 1)) * v / norm);
  }
  q->ofdm = srslte_hip_ofdm_create((int)P, 1, 0);
  q->dmrs = srslte_hip_chest_ul_create(cfg->cell_id, P, 1, &cfg->dmrs_cfg);
  bool ok = q->ofdm && q->dmrs && srslte_hip_ofdm_set_normalize(q->ofdm, 1) == SRSLTE_SUCCESS &&
            srslte_hip_ofdm_set_freq_shift(q->ofdm, 0.5f) == SRSLTE_SUCCESS; // ue_ul.c:63-64
  if (ok) { // srslte_sequence_pusch (sequences.c:65-67)
    std::vector<uint32_t> scr((size_t)10 * scr_words, 0);
    std::vector<uint8_t>  c;
    for (uint32_t sf = 0; sf < 10; sf++) {
      lte_gold_sequence(((uint32_t)cfg->rnti << 14) + (sf << 9) + cfg->cell_id, nbits, c);
      for (uint32_t i = 0; i < nbits; i++) scr[(size_t)sf * scr_words + (i >> 5)] |= (uint32_t)c[i] << (i & 31);
    }
    ok = upload(&q->d_scr, scr) == SRSLTE_SUCCESS;
  }
  if (ok) { // rate matching, rv 0 (rm_turbo.c:100-158): coded bit of each circular-buffer position, addressed in the encoder's byte streams
    std::vector<uint32_t> t;
    lte_rm_rx_table(K, 0, t);
    for (auto& v : t) {
      const uint32_t p = v / 3, s = v % 3;
      v = s == 0 ? (p < K ? p : (1u << 30) | (p - K)) : (2u << 30) | (s == 1 ? p : K + 4 + p);
    }
    ok = upload(&q->d_rm, t) == SRSLTE_SUCCESS;
  }
  if (ok && cfg->cqi_len > 11) { // srslte_rm_conv_tx (rm_conv.c:44-89): the sub-block interleaved streams read circularly, dummies skipped
    static const uint8_t perm[32] = {1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31, 0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30};
    const int            F = (int)cfg->cqi_len + 8, nrows = (F - 1) / 32 + 1, K_p = nrows * 32, ndummy = K_p - F, Q = g.Qp_cqi * (int)Qm;
    std::vector<int>      w;
    for (int st = 0; st < 3; st++) {
      for (int j = 0; j < 32; j++) {
        for (int i = 0; i < nrows; i++) {
          const int pos = i * 32 + perm[j];
          if (pos >= ndummy) w.push_back((pos - ndummy) * 3 + st);
        }
      }
    }
    std::vector<uint16_t> t((size_t)(Q > 0 ? Q : 1));
    for (int i = 0; i < Q; i++) t[i] = (uint16_t)w[(size_t)i % w.size()];
    ok = upload(&q->d_cqi_rm, t) == SRSLTE_SUCCESS;
  }
  if (ok && cfg->cqi_len) ok = hipMalloc((void**)&q->d_qcqi, (size_t)g.cqi_stride * B + 16) == hipSuccess;
  const size_t glen = (size_t)14 * 12 * P;
  ok = ok && hipMalloc((void**)&q->d_tbcrc, sizeof(uint32_t) * B) == hipSuccess &&
       hipMalloc((void**)&q->d_cb, (size_t)g.cb_stride * B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_parity, (size_t)g.par_stride * B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_sys_tail, (size_t)B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_d, sizeof(cf32) * (size_t)nof_re * B) == hipSuccess &&
       hipMalloc((void**)&q->d_z, sizeof(cf32) * (size_t)nof_re * B) == hipSuccess &&
       hipMalloc((void**)&q->d_grid, sizeof(cf32) * glen * B) == hipSuccess;
  if (!ok) {
    hip_log("[srslte_hip] ul_tx: initialisation failed\n");
    srslte_hip_ul_tx_destroy(q);
    return nullptr;
  }
  return q;
}

extern "C" const void* srslte_hip_ul_tx_debug_buffer(const srslte_hip_ul_tx_t* q, int which)
{
  if (!q) return nullptr;
  switch (which) {
    case 0: return q->d_cb;
    case 1: return q->d_parity;
    case 2: return q->d_d;
    case 3: return q->d_z;
    case 4: return q->d_grid;
    case 5: return q->d_tbcrc;
  }
  return nullptr;
}

extern "C" int srslte_hip_ul_tx_batch(srslte_hip_ul_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, uint32_t tti0, uint32_t nof_sf, void* d_iq,
                                      void* stream)
{
  if (q && (q->cfg.ack_len || q->cfg.ri_len || q->cfg.cqi_len)) return SRSLTE_ERROR_INVALID_INPUTS; // UCI configured: the values come through _batch_ack / _batch_uci / _batch_uci_cqi
  return srslte_hip_ul_tx_batch_ack(q, d_tb, tb_stride, nullptr, tti0, nof_sf, d_iq, stream);
}

extern "C" int srslte_hip_ul_tx_batch_ack(srslte_hip_ul_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, const uint8_t* d_ack, uint32_t tti0,
                                          uint32_t nof_sf, void* d_iq, void* stream)
{
  return srslte_hip_ul_tx_batch_uci(q, d_tb, tb_stride, d_ack, nullptr, tti0, nof_sf, d_iq, stream);
}

extern "C" int srslte_hip_ul_tx_batch_uci(srslte_hip_ul_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, const uint8_t* d_ack, const uint8_t* d_ri,
                                          uint32_t tti0, uint32_t nof_sf, void* d_iq, void* stream)
{
  return srslte_hip_ul_tx_batch_uci_cqi(q, d_tb, tb_stride, d_ack, d_ri, nullptr, tti0, nof_sf, d_iq, stream);
}

extern "C" int srslte_hip_ul_tx_batch_uci_cqi(srslte_hip_ul_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, const uint8_t* d_ack, const uint8_t* d_ri,
                                              const uint8_t* d_cqi, uint32_t tti0, uint32_t nof_sf, void* d_iq, void* stream)
{
  return srslte_hip_ul_tx_batch_rv(q, d_tb, tb_stride, d_ack, d_ri, d_cqi, 0, tti0, nof_sf, d_iq, stream);
}

// The same with a redundancy version (srslte_pusch_grant_t.tb.rv -> srslte_ulsch_encode -> srslte_rm_turbo_tx_lut's k0, rm_turbo.c:100-158):
// what a retransmission sends. Everything else - UCI multiplexing, interleaver, scrambling - does not depend on it.
extern "C" int srslte_hip_ul_tx_batch_rv(srslte_hip_ul_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, const uint8_t* d_ack, const uint8_t* d_ri,
                                         const uint8_t* d_cqi, uint32_t rv, uint32_t tti0, uint32_t nof_sf, void* d_iq, void* stream)
{
  if (!q || !d_tb || !d_iq || rv > 3 || nof_sf > q->cfg.max_batch || tb_stride < q->cfg.tbs / 8) return SRSLTE_ERROR_INVALID_INPUTS;
  if ((q->cfg.ack_len != 0) != (d_ack != nullptr) || (q->cfg.ri_len != 0) != (d_ri != nullptr) || (q->cfg.cqi_len != 0) != (d_cqi != nullptr))
    return SRSLTE_ERROR_INVALID_INPUTS;
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  hipStream_t st = (hipStream_t)stream;
  const void* d_r = nullptr;
  if (int r = chest_ul_dmrs_table(q->dmrs, q->cfg.L_prb, q->cfg.n_dmrs, &d_r)) return r;
  const uint32_t* d_rm = q->d_rm;
  if (rv) {
    if (!q->d_rm_rv[rv]) {
      const uint32_t        K = q->seg.K1;
      std::vector<uint32_t> t;
      lte_rm_rx_table(K, rv, t);
      for (auto& v : t) {
        const uint32_t p = v / 3, s = v % 3;
        v = s == 0 ? (p < K ? p : (1u << 30) | (p - K)) : (2u << 30) | (s == 1 ? p : K + 4 + p);
      }
      if (int r = upload(&q->d_rm_rv[rv], t)) return r;
    }
    d_rm = q->d_rm_rv[rv];
  }
  PuschTxGeom g = q->g;
  g.tti0        = (int)tti0;
  g.tb_stride   = (int)tb_stride;
  g.ack_bits    = d_ack;
  g.ri_bits     = d_ri;
  g.q_cqi       = d_cqi ? q->d_qcqi : nullptr;
  if (d_cqi) {
    hipLaunchKernelGGL(pusch_cqi_encode_kernel, dim3(nof_sf), dim3(256), 0, st, d_cqi, q->d_qcqi, g.cqi_stride, g.Qp_cqi * g.Qm, (int)q->cfg.cqi_len,
                       (const uint16_t*)q->d_cqi_rm);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(pusch_tx_tbcrc_kernel, dim3(nof_sf), dim3(256), 0, st, d_tb, q->d_tbcrc, g);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(pusch_tx_seg_kernel, dim3(g.C, nof_sf), dim3(256), 0, st, d_tb, (const uint32_t*)q->d_tbcrc, q->d_cb, g);
  LAUNCH_CHECK();
  int r = srslte_hip_tcod_encode_bytes_batch(q->d_cb, (uint32_t)g.cb_stride, q->d_parity, (uint32_t)g.par_stride, q->d_sys_tail, (uint32_t)g.K,
                                             nof_sf * (uint32_t)g.C, stream);
  if (r) return r;
  hipLaunchKernelGGL(pusch_tx_mod_kernel, dim3(ceil_div(g.M_sc, 256), g.nsymb, nof_sf), dim3(256), 0, st, (const uint8_t*)q->d_cb,
                     (const uint8_t*)q->d_parity, (const uint8_t*)q->d_sys_tail, d_rm, (const uint32_t*)q->d_scr, q->d_d, g);
  LAUNCH_CHECK();
  r = srslte_hip_dft_precoding_batch(q->d_d, q->d_z, q->cfg.L_prb, g.nsymb * nof_sf, 1, stream); // srslte_dft_precoding_init_tx: forward, 1/sqrt(N)
  if (r) return r;
  hipLaunchKernelGGL(pusch_tx_map_kernel, dim3(ceil_div(g.cell_nre, 256), 14, nof_sf), dim3(256), 0, st, (const cf32*)q->d_z, (const cf32*)d_r,
                     q->d_grid, g);
  LAUNCH_CHECK();
  return srslte_hip_ofdm_tx_sf_batch(q->ofdm, q->d_grid, d_iq, (int)nof_sf, stream);
}

// ====================================================================================================================
// PDSCH transmit pipeline (eNB side): srslte_pdsch_encode (pdsch.c:1059-1185: DL-SCH coding sch.c:183-297 with the
// Qm * N_L block split :549-575, scrambling, modulation, layer mapping + SFBC precoding, RE mapping) + CRS (srslte_refsignal_cs_put_sf,
// refsignal_dl.c:253-272) + srslte_ofdm_tx_sf with 1/sqrt(N) (enb_dl.c:56-62). One codeword, TM1 or 2-port TM2, full-band grant.
// Re-uses the PUSCH transmit kernels for CRC attachment / segmentation and the byte-stream turbo encoder.
// ====================================================================================================================
namespace {

struct PdschTxGeom {
  SfClass cls[3];
  const int32_t* src[3][4]; // per subframe class and port: grid RE -> >= 0 index into the port's symbol stream, -1 zero, <= -2 CRS pilot -(v + 2)
  int   grid_len, max_re, Qm, Nl, nof_ports, tti0, scr_words, C, K, cb_stride, par_stride, rm_len;
  float lvl[16], gain; // constellation levels of one axis; rho_a (TM1) or rho_a / sqrt(2) (TM2)
};

// grid = (ceil(max_re / (256 * G)), nof_sf), G = nof_ports: one thread per precoding group (one symbol for TM1, the SFBC pair 2i, 2i+1 for
// 2 ports, four symbols for 4 ports). Bit e of a code block = coded bit rm[e mod (3K+12)] in the encoder's byte streams, as in
// pusch_tx_mod_kernel; the block split counts in units of Qm * N_L bits (N_L = 2 with transmit diversity). y: [nof_sf][nof_ports][max_re].
// One transport block: nre symbols in precoding groups of G = nof_ports; cs: its scrambling bits; cb0: its first code-block slot; C / Qm / rm_len /
// lvl: of ITS segmentation and modulation; y0: its [nof_ports][max_re] symbol streams.
__device__ __forceinline__ void pdsch_tx_mod_body(const uint8_t* __restrict__ cb, const uint8_t* __restrict__ parity, const uint8_t* __restrict__ sys_tail,
                                                  const uint32_t* __restrict__ rm, const uint32_t* __restrict__ cs, cf32* __restrict__ y0, const PdschTxGeom& g,
                                                  int grp, int nre, int cb0, int C, int Qm, int rm_len, const float* __restrict__ lvl)
{
  const int G = g.nof_ports, Gp = nre / g.Nl; // Gp = G' of 36.212 5.1.4.1.2
  if (grp * G >= nre) return;
  const int QmL = Qm * g.Nl, gamma = Gp % C, lo = Gp / C, C_lo = C - gamma; // blocks 0..C_lo-1 carry lo units, the rest lo + 1 (sch.c:232-236)
  cf32            d[4];
  for (int t = 0; t < G; t++) {
    const int i = grp * G + t, u = i / g.Nl; // symbol, split unit
    int       r, e0;
    if (u < C_lo * lo) {
      r  = u / lo;
      e0 = (u - r * lo) * QmL;
    } else {
      const int v = u - C_lo * lo;
      r           = C_lo + v / (lo + 1);
      e0          = (v % (lo + 1)) * QmL;
    }
    e0 += (i % g.Nl) * Qm;
    const size_t   cbi = (size_t)cb0 + r;
    const uint8_t *xb = cb + cbi * g.cb_stride, *pb = parity + cbi * g.par_stride;
    const int      q0 = i * Qm;
    int            re = 0, im = 0;
    for (int b = 0; b < Qm; b++) {
      const uint32_t src = rm[(e0 + b) % rm_len], pos = src & 0x3fffffffu;
      const uint8_t  byte = (src >> 30) == 0 ? xb[pos >> 3] : ((src >> 30) == 1 ? sys_tail[cbi] : pb[pos >> 3]);
      int            bit  = (byte >> (7 - (pos & 7))) & 1;
      bit ^= (cs[(q0 + b) >> 5] >> ((q0 + b) & 31)) & 1;
      if (b & 1) im = (im << 1) | bit;
      else re = (re << 1) | bit;
    }
    d[t] = make_float2(lvl[re] * g.gain, lvl[im] * g.gain);
  }
  const cf32 z  = make_float2(0.f, 0.f);
  if (G == 1) {
    y0[grp] = d[0];
  } else if (G == 2) { // srslte_precoding_diversity, 2 ports (precoding.c:1851-1861): y0 = x0, x1; y1 = -x1*, x0*
    cf32* y1        = y0 + g.max_re;
    y0[2 * grp]     = d[0];
    y0[2 * grp + 1] = d[1];
    y1[2 * grp]     = make_float2(-d[1].x, d[1].y);
    y1[2 * grp + 1] = make_float2(d[0].x, -d[0].y);
  } else { // 4 ports (precoding.c:1862-1890): ports 0/2 on sub-carriers 4i, 4i+1, ports 1/3 on 4i+2, 4i+3, the others silent
    cf32 *y1 = y0 + g.max_re, *y2 = y1 + g.max_re, *y3 = y2 + g.max_re;
    const int k = 4 * grp;
    y0[k] = d[0];     y1[k] = z;        y2[k] = make_float2(-d[1].x, d[1].y);     y3[k] = z;
    y0[k + 1] = d[1]; y1[k + 1] = z;    y2[k + 1] = make_float2(d[0].x, -d[0].y); y3[k + 1] = z;
    y0[k + 2] = z;    y1[k + 2] = d[2]; y2[k + 2] = z; y3[k + 2] = make_float2(-d[3].x, d[3].y);
    y0[k + 3] = z;    y1[k + 3] = d[3]; y2[k + 3] = z; y3[k + 3] = make_float2(d[2].x, -d[2].y);
  }
}


__global__ __launch_bounds__(256) void pdsch_tx_mod_kernel(const uint8_t* __restrict__ cb, const uint8_t* __restrict__ parity,
                                                           const uint8_t* __restrict__ sys_tail, const uint32_t* __restrict__ rm,
                                                           const uint32_t* __restrict__ scr, cf32* __restrict__ y, PdschTxGeom g)
{
  const int sf = blockIdx.y, sf_idx = (g.tti0 + sf) % 10, nre = g.cls[sf_class(sf_idx)].nof_re, grp = blockIdx.x * blockDim.x + threadIdx.x;
  if (grp * g.nof_ports >= nre) return;
  pdsch_tx_mod_body(cb, parity, sys_tail, rm, scr + (size_t)sf_idx * g.scr_words, y + ((size_t)sf * g.nof_ports) * g.max_re, g, grp, nre, sf * g.C, g.C, g.Qm,
                    g.rm_len, g.lvl);
}

// ---- per-PDSCH grants on the transmit side (srslte_hip_dl_tx_batch_grants): PDSCH p of a call has its own allocation, RNTI, modulation, transport
// block and redundancy version; several may share a subframe's grid
struct TxDesc {
  int             row, sf;             // row of the caller's d_tb; subframe of the batch
  int             tbs, C, K, rlenB;    // segmentation of its transport block
  int             cb0;                 // its first code-block slot (slots have the strides of the largest block size)
  int             nre, mod, Qm;
  const uint32_t* rm;                  // rate-matching table of (K, rv)
};
struct TxLevels { float v[5][16]; };   // constellation levels of one axis per srslte_mod_t

// grid = nof_pdsch: CRC24A of each transport block
__global__ __launch_bounds__(256) void tx_tbcrc_grants_kernel(const uint8_t* __restrict__ tb, int tb_stride, const TxDesc* __restrict__ desc,
                                                              uint32_t* __restrict__ crc_out)
{
  __shared__ uint32_t tab[256], red[256];
  const uint8_t*      x = tb + (size_t)desc[blockIdx.x].row * tb_stride;
  const uint32_t      c = block_crc24([&](int i) { return (uint32_t)x[i]; }, desc[blockIdx.x].tbs / 8, 0x1864CFBu, tab, red);
  if (threadIdx.x == 0) crc_out[blockIdx.x] = c;
}

// grid = (Cmax, nof_pdsch): segmentation + CRC24B as pusch_tx_seg_kernel, per descriptor
__global__ __launch_bounds__(256) void tx_seg_grants_kernel(const uint8_t* __restrict__ tb, int tb_stride, const uint32_t* __restrict__ tbcrc,
                                                            const TxDesc* __restrict__ desc, uint8_t* __restrict__ cb, int cb_stride)
{
  __shared__ uint32_t tab[256], red[256];
  __shared__ uint8_t  xs[768];
  const TxDesc&       d = desc[blockIdx.y];
  const int           r = blockIdx.x, t = threadIdx.x, tbB = d.tbs / 8, rlenB = d.rlenB;
  if (r >= d.C) return;
  const uint8_t* x   = tb + (size_t)d.row * tb_stride;
  const uint32_t crc = tbcrc[blockIdx.y];
  uint8_t*       out = cb + ((size_t)d.cb0 + r) * cb_stride;
  for (int i = t; i < rlenB; i += 256) {
    const int j = r * rlenB + i;
    xs[i]       = j < tbB ? x[j] : (uint8_t)(crc >> (8 * (2 - (j - tbB))));
  }
  __syncthreads();
  for (int i = t; i < rlenB; i += 256) out[i] = xs[i];
  if (d.C > 1) {
    const uint32_t c = block_crc24([&](int i) { return (uint32_t)xs[i]; }, rlenB, 0x1800063u, tab, red);
    if (t < 3) out[rlenB + t] = (uint8_t)(c >> (8 * (2 - t)));
  }
}

// grid = (ceil(max_re / (256 * G)), nof_pdsch)
__global__ __launch_bounds__(256) void pdsch_tx_mod_grants_kernel(const uint8_t* __restrict__ cb, const uint8_t* __restrict__ parity,
                                                                  const uint8_t* __restrict__ sys_tail, const uint32_t* __restrict__ scr, int scr_words,
                                                                  cf32* __restrict__ y, const TxDesc* __restrict__ desc, TxLevels lv, PdschTxGeom g)
{
  const TxDesc& d   = desc[blockIdx.y];
  const int     grp = blockIdx.x * blockDim.x + threadIdx.x;
  if (grp * g.nof_ports >= d.nre) return;
  pdsch_tx_mod_body(cb, parity, sys_tail, d.rm, scr + (size_t)blockIdx.y * scr_words, y + ((size_t)blockIdx.y * g.nof_ports) * g.max_re, g, grp, d.nre, d.cb0,
                    d.C, d.Qm, 3 * d.K + 12, lv.v[d.mod]);
}

// grid = (ceil(max_re / 256), nof_pdsch * nof_ports): the symbols of PDSCH p, port by port, onto the REs of its list in its subframe's grids
__global__ __launch_bounds__(256) void pdsch_tx_scatter_kernel(const cf32* __restrict__ y, const uint32_t* __restrict__ relist, cf32* __restrict__ grid,
                                                               const TxDesc* __restrict__ desc, int max_re, int grid_len, int nof_ports)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y / nof_ports, port = blockIdx.y - p * nof_ports;
  if (i >= desc[p].nre) return;
  grid[((size_t)desc[p].sf * nof_ports + port) * grid_len + relist[(size_t)p * max_re + i]] = y[((size_t)p * nof_ports + port) * max_re + i];
}

// grid = (ceil(grid_len/256), nof_sf * nof_ports): the resource grid of one port: PDSCH symbols, this port's CRS, zero elsewhere
__global__ __launch_bounds__(256) void pdsch_tx_map_kernel(const cf32* __restrict__ y, const cf32* __restrict__ pilots, cf32* __restrict__ grid,
                                                           int nref4 /* 4 * 2 * nof_prb */, PdschTxGeom g)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x, sp = blockIdx.y, sf = sp / g.nof_ports, port = sp - sf * g.nof_ports;
  if (k >= g.grid_len) return;
  const int sf_idx = (g.tti0 + sf) % 10, v = g.src[sf_class(sf_idx)][port][k];
  cf32      o = make_float2(0.f, 0.f);
  if (v >= 0) o = y[(size_t)sp * g.max_re + v];
  else if (v <= -2) { // ports 0/1: [10][4][nref]; ports 2/3: [10][2][nref] behind them (chest.hip)
    o = port < 2 ? pilots[(size_t)sf_idx * nref4 + (-(v + 2))] : pilots[(size_t)10 * nref4 + (size_t)sf_idx * (nref4 / 2) + (-(v + 2))];
  }
  grid[(size_t)sp * g.grid_len + k] = o;
}

} // namespace

struct srslte_hip_dl_tx {
  srslte_hip_dl_tx_cfg_t cfg;
  srslte_hip_ofdm_t*     ofdm;
  srslte_hip_chest_dl_t* crs; // for its CRS table
  srslte_hip_cbsegm_t    seg;
  PuschTxGeom            cg; // CRC attachment / segmentation geometry (shared kernels)
  PdschTxGeom            g;
  uint32_t *             d_scr, *d_rm[4], *d_tbcrc, *d_idx[3];
  int32_t*               d_src[3][4];
  uint8_t *              d_cb, *d_parity, *d_sys_tail;
  cf32 *                 d_y, *d_grid;
  struct TxGrantsState*  gs; // srslte_hip_dl_tx_batch_grants: created on first use
};

// Device / host resources of the per-PDSCH grants mode of the transmit pipeline
struct TxGrantsState {
  uint32_t  V, Cmax, max_re, words, cb_stride, par_stride;
  uint32_t *d_relist, *d_scr, *d_basis, *d_tbcrc;
  uint8_t * d_cb, *d_parity, *d_sys_tail, *d_desc;
  cf32*     d_y;
  int32_t*  d_crs_src[4]; // per port: grid RE -> -1 (zero) or the CRS pilot -(v + 2), for pdsch_tx_map_kernel as the grid initialiser
  size_t    desc_bytes;
  uint8_t*   h_pin[4];
  hipEvent_t h_ev[4];
  bool       h_used[4];
  uint32_t   h_slot;
  TxLevels   lv;
  std::map<std::pair<uint32_t, uint32_t>, uint32_t*> rm_tbl; // (K, rv) -> rate-matching table over the encoder's byte streams
};

static void tx_grants_free(TxGrantsState* g)
{
  if (!g) return;
  void* gb[] = {g->d_relist, g->d_scr, g->d_basis, g->d_tbcrc, g->d_cb, g->d_parity, g->d_sys_tail, g->d_desc, g->d_y,
                g->d_crs_src[0], g->d_crs_src[1], g->d_crs_src[2], g->d_crs_src[3]};
  for (void* b : gb) {
    if (b) (void)hipFree(b);
  }
  for (auto& kv : g->rm_tbl) (void)hipFree(kv.second);
  for (int i = 0; i < 4; i++) {
    if (g->h_pin[i]) {
      (void)hipHostFree(g->h_pin[i]);
      (void)hipEventDestroy(g->h_ev[i]);
    }
  }
  delete g;
}

extern "C" void srslte_hip_dl_tx_destroy(srslte_hip_dl_tx_t* q)
{
  if (!q) return;
  srslte_hip_ofdm_destroy(q->ofdm);
  srslte_hip_chest_dl_destroy(q->crs);
  void* bufs[] = {q->d_scr, q->d_rm[0], q->d_rm[1], q->d_rm[2], q->d_rm[3], q->d_tbcrc, q->d_idx[0], q->d_idx[1], q->d_idx[2],
                  q->d_cb, q->d_parity, q->d_sys_tail, q->d_y, q->d_grid};
  for (void* b : bufs) {
    if (b) (void)hipFree(b);
  }
  for (auto& cls : q->d_src) {
    for (int32_t* b : cls) {
      if (b) (void)hipFree(b);
    }
  }
  tx_grants_free(q->gs);
  delete q;
}

static int dl_tx_rm_table(srslte_hip_dl_tx_t* q, uint32_t rv)
{ // rate matching (rm_turbo.c:100-158): coded bit of each circular-buffer position read from k0(rv), addressed in the encoder's byte streams
  const uint32_t        K = q->seg.K1;
  std::vector<uint32_t> t;
  lte_rm_rx_table(K, rv, t);
  for (auto& v : t) {
    const uint32_t p = v / 3, s = v % 3;
    v = s == 0 ? (p < K ? p : (1u << 30) | (p - K)) : (2u << 30) | (s == 1 ? p : K + 4 + p);
  }
  return upload(&q->d_rm[rv], t);
}

extern "C" srslte_hip_dl_tx_t* srslte_hip_dl_tx_create(const srslte_hip_dl_tx_cfg_t* cfg)
{
  if (!cfg || cfg->max_batch == 0 || cfg->mod < 1 || cfg->mod > 4 || cfg->nof_ports > 4 || cfg->nof_ports == 3 || cfg->nof_prb < 6 || cfg->nof_prb > 110) {
    hip_log("[srslte_hip] dl_tx: invalid configuration\n");
    return nullptr;
  }
  auto* q = new srslte_hip_dl_tx();
  memset(q, 0, sizeof(*q));
  q->cfg = *cfg;
  if (srslte_hip_cbsegm(&q->seg, cfg->tbs) || q->seg.F || q->seg.C2 || (cfg->tbs % 8)) {
    hip_log("[srslte_hip] dl_tx: TBS %u needs filler bits or two code-block sizes; not supported on device yet\n", cfg->tbs);
    delete q;
    return nullptr;
  }
  const uint32_t P = cfg->nof_prb, nre = 12 * P, B = cfg->max_batch, C = q->seg.C, K = q->seg.K1, Qm = 2 * (uint32_t)cfg->mod;
  const uint32_t lstart = cfg->cfi + (P < 10 ? 1 : 0), npt = cfg->nof_ports ? cfg->nof_ports : 1, glen = 14 * nre;
  q->ofdm = srslte_hip_ofdm_create((int)P, 1, 0);
  q->crs  = srslte_hip_chest_dl_create(cfg->cell_id, P, npt, 1);
  bool ok = q->ofdm && q->crs && srslte_hip_ofdm_set_normalize(q->ofdm, 1) == SRSLTE_SUCCESS; // enb_dl.c:61
  uint32_t       max_re    = 0;
  const uint32_t rep_sf[3] = {0, 5, 1};
  PdschTxGeom&   g = q->g;
  for (int c = 0; c < 3 && ok; c++) {
    std::vector<uint32_t> idx;
    pdsch_re_indices(cfg->cell_id, P, npt, rep_sf[c], lstart, idx);
    g.cls[c].nof_re = (int)idx.size();
    max_re          = idx.size() > max_re ? (uint32_t)idx.size() : max_re;
    ok              = upload(&q->d_idx[c], idx) == SRSLTE_SUCCESS;
    g.cls[c].idx    = q->d_idx[c];
    for (uint32_t port = 0; port < npt && ok; port++) {
      std::vector<int32_t> src(glen, -1);
      for (size_t i = 0; i < idx.size(); i++) src[idx[i]] = (int32_t)i;
      for (int l = 0; l < (port < 2 ? 4 : 2); l++) { // srslte_refsignal_cs_put_sf (refsignal_dl.c:253-272): ports 0/1 symbols 0, 4, 7, 11; ports 2/3 symbols 1, 8
        const uint32_t sym = port >= 2 ? 1 + 7 * l : ((l & 1) ? (l / 2 + 1) * 7 - 3 : (l / 2) * 7), fidx = ((((l + port) & 1) ? 3 : 0) + cfg->cell_id % 6) % 6;
        for (uint32_t i = 0; i < 2 * P; i++) src[sym * nre + fidx + 6 * i] = -(int32_t)(l * 2 * P + i) - 2;
      }
      ok             = upload(&q->d_src[c][port], src) == SRSLTE_SUCCESS;
      g.src[c][port] = q->d_src[c][port];
    }
  }
  const uint32_t max_bits = max_re * Qm, scr_words = (max_bits + 31) / 32;
  if (ok) { // srslte_sequence_pdsch (sequences.c:58-60), codeword 0
    std::vector<uint32_t> scr((size_t)10 * scr_words, 0);
    std::vector<uint8_t>  c;
    for (uint32_t sf = 0; sf < 10; sf++) {
      lte_gold_sequence(((uint32_t)cfg->rnti << 14) + (sf << 9) + cfg->cell_id, max_bits, c);
      for (uint32_t i = 0; i < max_bits; i++) scr[(size_t)sf * scr_words + (i >> 5)] |= (uint32_t)c[i] << (i & 31);
    }
    ok = upload(&q->d_scr, scr) == SRSLTE_SUCCESS;
  }
  ok = ok && dl_tx_rm_table(q, 0) == SRSLTE_SUCCESS;
  PuschTxGeom& cg = q->cg;
  cg.C = (int)C; cg.K = (int)K; cg.tbs = (int)cfg->tbs; cg.rlenB = (int)((C == 1 ? K : K - 24) / 8); cg.cb_stride = (int)((K / 8 + 15) & ~15u);
  cg.par_stride = (int)((K / 4 + 1 + 15) & ~15u);
  g.grid_len = (int)glen; g.max_re = (int)max_re; g.Qm = (int)Qm; g.Nl = npt > 1 ? 2 : 1; g.nof_ports = (int)npt; g.scr_words = (int)scr_words;
  g.C = (int)C; g.K = (int)K; g.cb_stride = cg.cb_stride; g.par_stride = cg.par_stride; g.rm_len = (int)(3 * K + 12);
  for (uint32_t idx = 0; idx < (1u << cfg->mod); idx++) { // 36.211 7.1.2-7.1.5, one axis (lte_tables.c:57-262)
    const int    nb = cfg->mod;
    double       v  = 1.0;
    for (int i = nb - 1; i >= 1; i--) v = (double)(1 << (nb - i)) - (1 - 2 * (int)((idx >> (nb - 1 - i)) & 1)) * v;
    const double norm = nb == 1 ? sqrt(2.0) : (nb == 2 ? sqrt(10.0) : (nb == 3 ? sqrt(42.0) : sqrt(170.0)));
    g.lvl[idx]        = (float)((1 - 2 * (int)((idx >> (nb - 1)) & 1)) * v / norm);
  }
  const float rho_a = powf(10.0f, cfg->p_a / 20.0f) * (npt == 1 ? 1.0f : sqrtf(2.0f)); // pdsch.c:525
  g.gain            = npt == 1 ? rho_a : rho_a / sqrtf(2.0f);                          // precoding.c:1859-1860
  ok = ok && hipMalloc((void**)&q->d_tbcrc, sizeof(uint32_t) * B) == hipSuccess &&
       hipMalloc((void**)&q->d_cb, (size_t)cg.cb_stride * B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_parity, (size_t)cg.par_stride * B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_sys_tail, (size_t)B * C) == hipSuccess &&
       hipMalloc((void**)&q->d_y, sizeof(cf32) * (size_t)max_re * B * npt) == hipSuccess &&
       hipMalloc((void**)&q->d_grid, sizeof(cf32) * (size_t)glen * B * npt) == hipSuccess;
  if (!ok) {
    hip_log("[srslte_hip] dl_tx: initialisation failed\n");
    srslte_hip_dl_tx_destroy(q);
    return nullptr;
  }
  return q;
}

extern "C" const void* srslte_hip_dl_tx_debug_buffer(const srslte_hip_dl_tx_t* q, int which)
{
  if (!q) return nullptr;
  switch (which) {
    case 0: return q->d_cb;
    case 1: return q->d_parity;
    case 2: return q->d_y;
    case 3: return q->d_grid;
  }
  return nullptr;
}

extern "C" int srslte_hip_dl_tx_batch(srslte_hip_dl_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, uint32_t tti0, uint32_t nof_sf, uint32_t rv,
                                      void* d_iq, void* stream)
{
  if (!q || !d_tb || !d_iq || nof_sf > q->cfg.max_batch || tb_stride < q->cfg.tbs / 8 || rv > 3) return SRSLTE_ERROR_INVALID_INPUTS;
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  if (!q->d_rm[rv]) {
    if (int r = dl_tx_rm_table(q, rv)) return r;
  }
  hipStream_t st = (hipStream_t)stream;
  PuschTxGeom cg = q->cg;
  cg.tb_stride   = (int)tb_stride;
  hipLaunchKernelGGL(pusch_tx_tbcrc_kernel, dim3(nof_sf), dim3(256), 0, st, d_tb, q->d_tbcrc, cg);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(pusch_tx_seg_kernel, dim3(cg.C, nof_sf), dim3(256), 0, st, d_tb, (const uint32_t*)q->d_tbcrc, q->d_cb, cg);
  LAUNCH_CHECK();
  int r = srslte_hip_tcod_encode_bytes_batch(q->d_cb, (uint32_t)cg.cb_stride, q->d_parity, (uint32_t)cg.par_stride, q->d_sys_tail, (uint32_t)cg.K,
                                             nof_sf * (uint32_t)cg.C, stream);
  if (r) return r;
  PdschTxGeom g = q->g;
  g.tti0        = (int)tti0;
  hipLaunchKernelGGL(pdsch_tx_mod_kernel, dim3(ceil_div(g.max_re / g.nof_ports, 256), nof_sf), dim3(256), 0, st, (const uint8_t*)q->d_cb,
                     (const uint8_t*)q->d_parity, (const uint8_t*)q->d_sys_tail, (const uint32_t*)q->d_rm[rv], (const uint32_t*)q->d_scr, q->d_y, g);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(pdsch_tx_map_kernel, dim3(ceil_div(g.grid_len, 256), nof_sf * g.nof_ports), dim3(256), 0, st, (const cf32*)q->d_y,
                     (const cf32*)srslte_hip_chest_dl_pilots(q->crs), q->d_grid, 8 * (int)q->cfg.nof_prb, g);
  LAUNCH_CHECK();
  return srslte_hip_ofdm_tx_sf_batch(q->ofdm, q->d_grid, d_iq, (int)nof_sf * g.nof_ports, stream);
}

static int dl_tx_grants_init(srslte_hip_dl_tx_t* q, uint32_t V)
{
  const uint32_t P = q->cfg.nof_prb, cell_id = q->cfg.cell_id;
  const int      npt = q->g.nof_ports;
  auto*          g = new TxGrantsState(); // value-initialised: every pointer and flag starts null / false
  q->gs         = g;
  g->V          = V;
  g->Cmax       = q->seg.C;
  g->max_re     = 14 * 12 * P;
  g->words      = (g->max_re * 8 + 31) / 32 + 2;
  g->cb_stride  = (6144 / 8 + 15) & ~15u;
  g->par_stride = (6144 / 4 + 1 + 15) & ~15u;
  const size_t nblk = (size_t)V * g->Cmax;
  g->desc_bytes     = (sizeof(GrantDev) + sizeof(TxDesc)) * V;
  for (int i = 0; i < 4; i++) {
    HIP_TRY(hipEventCreateWithFlags(&g->h_ev[i], hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void**)&g->h_pin[i], g->desc_bytes));
  }
  if (gold_basis_upload(g->words, &g->d_basis)) return SRSLTE_ERROR;
  HIP_TRY(hipMalloc((void**)&g->d_relist, sizeof(uint32_t) * (size_t)g->max_re * V));
  HIP_TRY(hipMalloc((void**)&g->d_scr, sizeof(uint32_t) * (size_t)g->words * V));
  HIP_TRY(hipMalloc((void**)&g->d_tbcrc, sizeof(uint32_t) * V));
  HIP_TRY(hipMalloc((void**)&g->d_cb, (size_t)g->cb_stride * nblk));
  HIP_TRY(hipMalloc((void**)&g->d_parity, (size_t)g->par_stride * nblk));
  HIP_TRY(hipMalloc((void**)&g->d_sys_tail, nblk));
  HIP_TRY(hipMalloc((void**)&g->d_desc, g->desc_bytes));
  HIP_TRY(hipMalloc((void**)&g->d_y, sizeof(cf32) * (size_t)g->max_re * V * npt));
  for (int port = 0; port < npt; port++) { // srslte_refsignal_cs_put_sf (refsignal_dl.c:253-272), as srslte_hip_dl_tx_create maps it
    std::vector<int32_t> src((size_t)14 * 12 * P, -1);
    for (int l = 0; l < (port < 2 ? 4 : 2); l++) {
      const uint32_t sym = port >= 2 ? 1 + 7 * l : ((l & 1) ? (l / 2 + 1) * 7 - 3 : (l / 2) * 7), fidx = ((((l + port) & 1) ? 3 : 0) + cell_id % 6) % 6;
      for (uint32_t i = 0; i < 2 * P; i++) src[sym * 12 * P + fidx + 6 * i] = -(int32_t)(l * 2 * P + i) - 2;
    }
    if (upload(&g->d_crs_src[port], src)) return SRSLTE_ERROR;
  }
  for (int mod = 1; mod <= 4; mod++) { // 36.211 7.1.2-7.1.5, one axis (lte_tables.c:57-262)
    for (uint32_t idx = 0; idx < (1u << mod); idx++) {
      double v = 1.0;
      for (int i = mod - 1; i >= 1; i--) v = (double)(1 << (mod - i)) - (1 - 2 * (int)((idx >> (mod - 1 - i)) & 1)) * v;
      const double norm = mod == 1 ? sqrt(2.0) : (mod == 2 ? sqrt(10.0) : (mod == 3 ? sqrt(42.0) : sqrt(170.0)));
      g->lv.v[mod][idx] = (float)((1 - 2 * (int)((idx >> (mod - 1)) & 1)) * v / norm);
    }
  }
  return SRSLTE_SUCCESS;
}

// Per-PDSCH grants on the transmit side: what an eNB sends in a run of TTIs - srslte_enb_dl_put_base once per TTI, then srslte_enb_dl_put_pdsch
// once per scheduled UE (enb_dl.c:330-398 -> srslte_pdsch_encode, pdsch.c:1059-1185), each with its own srslte_pdsch_grant_t, then
// srslte_enb_dl_gen_signal. grants[p]: the subframe of the batch, and a srslte_hip_dl_grant_t as the receive side takes it (PRB masks of both
// slots, modulation, transport block, redundancy version, RNTI, CFI; new_data is not used). Row p of d_tb is its transport block. The grids are
// initialised with the CRS of every port, each PDSCH's symbols go onto the REs pdsch_relist_kernel lists for its masks (srslte_pdsch_cp, put =
// true, including upstream's stale-offset rule); allocations that overlap within a subframe are the caller's error (which PDSCH wins an RE is
// not defined here; upstream the later put would). The object's
// cell, antenna ports (TM1 / transmit diversity), p_a apply; cfg.tbs bounds every grant's tbs, cfg.max_grants the number of PDSCHs per call.
extern "C" int srslte_hip_dl_tx_batch_grants(srslte_hip_dl_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, uint32_t tti0, uint32_t nof_sf,
                                             const srslte_hip_dl_tx_grant_t* grants, uint32_t nof_grants, void* d_iq, void* stream)
{
  if (!q || !d_tb || !d_iq || !grants || nof_sf > q->cfg.max_batch) return SRSLTE_ERROR_INVALID_INPUTS;
  const uint32_t V = q->cfg.max_grants ? q->cfg.max_grants : q->cfg.max_batch, P = q->cfg.nof_prb, cell_id = q->cfg.cell_id;
  const int      npt = q->g.nof_ports;
  if (nof_grants > V) return SRSLTE_ERROR_INVALID_INPUTS;
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  hipStream_t st = (hipStream_t)stream;
  if (!q->gs && dl_tx_grants_init(q, V)) { // a failed start leaves no half-made state behind
    tx_grants_free(q->gs);
    q->gs = nullptr;
    return SRSLTE_ERROR;
  }
  TxGrantsState* g  = q->gs;
  const uint32_t hs = g->h_slot++ & 3u;
  if (g->h_used[hs]) HIP_TRY(hipEventSynchronize(g->h_ev[hs]));
  auto* h_gr = reinterpret_cast<GrantDev*>(g->h_pin[hs]);
  auto* h_td = reinterpret_cast<TxDesc*>(h_gr + V);
  auto* d_gr = reinterpret_cast<GrantDev*>(g->d_desc);
  auto* d_td = reinterpret_cast<TxDesc*>(d_gr + V);
  // code-block slots in the order of the block length, so that the encoder runs once per length over neighbouring slots
  std::vector<uint32_t>            order(nof_grants);
  std::vector<srslte_hip_cbsegm_t> segs(nof_grants);
  uint32_t                         max_nre = 0;
  for (uint32_t p = 0; p < nof_grants; p++) {
    const srslte_hip_dl_grant_t& gr = grants[p].grant;
    order[p] = p;
    if (grants[p].sf >= nof_sf || gr.mod < 1 || gr.mod > 4 || gr.rv > 3 || gr.cfi < 1 || gr.cfi > 3 || gr.tbs == 0 || gr.tbs > q->cfg.tbs || (gr.tbs % 8) ||
        tb_stride < gr.tbs / 8 || srslte_hip_cbsegm(&segs[p], gr.tbs) || segs[p].F || segs[p].C2 || segs[p].C > g->Cmax) {
      hip_log("[srslte_hip] dl_tx grants: entry %u: unsupported grant (subframe %u of %u, mod %d, tbs %u, rv %u, cfi %u)\n", p, grants[p].sf, nof_sf, gr.mod,
              gr.tbs, gr.rv, gr.cfi);
      return SRSLTE_ERROR_INVALID_INPUTS;
    }
  }
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return segs[a].K1 < segs[b].K1; });
  uint32_t cb0 = 0;
  for (uint32_t i = 0; i < nof_grants; i++) {
    const uint32_t               p  = order[i];
    const srslte_hip_dl_grant_t& gr = grants[p].grant;
    GrantDev&                    gd = h_gr[p];
    memset(&gd, 0, sizeof(gd));
    gd.sf_idx = (int)((tti0 + grants[p].sf) % 10); gd.lstart = (int)(gr.cfi + (P < 10 ? 1 : 0)); gd.rnti = gr.rnti;
    const uint32_t nre = pdsch_grant_dev(gr, P, cell_id, npt, gd);
    if (nre == 0 || (nre % (uint32_t)npt) || nre < segs[p].C * (uint32_t)q->g.Nl) {
      hip_log("[srslte_hip] dl_tx grants: entry %u: %u REs do not carry %u code blocks on %d ports\n", p, nre, segs[p].C, npt);
      return SRSLTE_ERROR_INVALID_INPUTS;
    }
    max_nre = nre > max_nre ? nre : max_nre;
    const uint32_t K  = segs[p].K1;
    auto           it = g->rm_tbl.find({K, gr.rv});
    if (it == g->rm_tbl.end()) { // rate matching (rm_turbo.c:100-158) addressed in the encoder's byte streams, as dl_tx_rm_table
      std::vector<uint32_t> t;
      lte_rm_rx_table(K, gr.rv, t);
      for (auto& v : t) {
        const uint32_t pos = v / 3, sidx = v % 3;
        v = sidx == 0 ? (pos < K ? pos : (1u << 30) | (pos - K)) : (2u << 30) | (sidx == 1 ? pos : K + 4 + pos);
      }
      uint32_t* d = nullptr;
      if (upload(&d, t)) return SRSLTE_ERROR;
      it = g->rm_tbl.emplace(std::make_pair(K, gr.rv), d).first;
    }
    TxDesc& td = h_td[p];
    td.row = (int)p; td.sf = (int)grants[p].sf; td.tbs = (int)gr.tbs; td.C = (int)segs[p].C; td.K = (int)K; td.rlenB = (int)((segs[p].C == 1 ? K : K - 24) / 8);
    td.cb0 = (int)cb0; td.nre = (int)nre; td.mod = gr.mod; td.Qm = 2 * gr.mod; td.rm = it->second;
    cb0 += segs[p].C;
  }
  HIP_TRY(hipMemcpyAsync(g->d_desc, g->h_pin[hs], g->desc_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(g->h_ev[hs], st));
  g->h_used[hs] = true;
  PdschTxGeom tg = q->g; // ports, N_L, gain; the grid initialiser's maps
  tg.max_re = (int)g->max_re;
  tg.cb_stride = (int)g->cb_stride; tg.par_stride = (int)g->par_stride; // the slots of this mode are spaced for the largest block length
  for (auto& c : tg.src) {
    for (int port = 0; port < 4; port++) c[port] = g->d_crs_src[port];
  }
  tg.tti0 = (int)tti0;
  hipLaunchKernelGGL(pdsch_tx_map_kernel, dim3(ceil_div(tg.grid_len, 256), nof_sf * npt), dim3(256), 0, st, (const cf32*)g->d_y,
                     (const cf32*)srslte_hip_chest_dl_pilots(q->crs), q->d_grid, 8 * (int)P, tg);
  LAUNCH_CHECK();
  if (nof_grants) {
    hipLaunchKernelGGL(pdsch_relist_kernel, dim3(nof_grants), dim3(RELIST_THREADS), 0, st, (const GrantDev*)d_gr, g->d_relist, (int)P, (int)cell_id,
                       (int)g->max_re, npt);
    hipLaunchKernelGGL(scr_gen_kernel, dim3(ceil_div((int)g->words, 256), nof_grants), dim3(256), 0, st, (const GrantDev*)d_gr, (const uint32_t*)g->d_basis,
                       g->d_scr, (int)g->words, (int)cell_id);
    hipLaunchKernelGGL(tx_tbcrc_grants_kernel, dim3(nof_grants), dim3(256), 0, st, d_tb, (int)tb_stride, (const TxDesc*)d_td, g->d_tbcrc);
    hipLaunchKernelGGL(tx_seg_grants_kernel, dim3(g->Cmax, nof_grants), dim3(256), 0, st, d_tb, (int)tb_stride, (const uint32_t*)g->d_tbcrc, (const TxDesc*)d_td,
                       g->d_cb, (int)g->cb_stride);
    LAUNCH_CHECK();
    for (uint32_t i = 0; i < nof_grants;) { // the encoder: runs of equal block length
      uint32_t j = i, n = 0;
      while (j < nof_grants && segs[order[j]].K1 == segs[order[i]].K1) n += segs[order[j++]].C;
      const size_t s0 = (size_t)h_td[order[i]].cb0;
      if (int r = srslte_hip_tcod_encode_bytes_batch(g->d_cb + s0 * g->cb_stride, g->cb_stride, g->d_parity + s0 * g->par_stride, g->par_stride,
                                                     g->d_sys_tail + s0, segs[order[i]].K1, n, stream))
        return r;
      i = j;
    }
    hipLaunchKernelGGL(pdsch_tx_mod_grants_kernel, dim3(ceil_div((int)max_nre / npt, 256), nof_grants), dim3(256), 0, st, (const uint8_t*)g->d_cb,
                       (const uint8_t*)g->d_parity, (const uint8_t*)g->d_sys_tail, (const uint32_t*)g->d_scr, (int)g->words, g->d_y, (const TxDesc*)d_td, g->lv, tg);
    hipLaunchKernelGGL(pdsch_tx_scatter_kernel, dim3(ceil_div((int)max_nre, 256), nof_grants * npt), dim3(256), 0, st, (const cf32*)g->d_y,
                       (const uint32_t*)g->d_relist, q->d_grid, (const TxDesc*)d_td, (int)g->max_re, tg.grid_len, npt);
    LAUNCH_CHECK();
  }
  return srslte_hip_ofdm_tx_sf_batch(q->ofdm, q->d_grid, d_iq, (int)nof_sf * npt, stream);
}

// ====================================================================================================================
// Per-PUSCH grants on the transmit side (srslte_hip_ul_tx_batch_grants): every PUSCH of a call has its own allocation, DMRS cyclic shift, RNTI,
// modulation, transport block, redundancy version and UCI - what srslte_ue_ul_encode sends TTI after TTI as the grants come in (ue_ul.c:300-340),
// and, with several PUSCHs on disjoint PRBs of one subframe, the composite signal of several UEs as an eNB receives it.
// ====================================================================================================================
namespace {

struct PuschTxDesc {
  int             M_sc, n_prb, n_prb1, zoff, syms_lo, C_lo, Qp_cqi, cqi_O, cqi_wlen, sf_idx;
  AckGeom         ack, ri;
  const uint16_t* cqi_w; // srslte_rm_conv_tx order of the report's coded bits (reports above 11 bits), read circularly
  const cf32*     dmrs;  // table of (L_prb, n_dmrs): [10][2][M_sc]
};

// grid = nof_pusch: the coded CQI report of each PUSCH that carries one (pusch_cqi_encode_kernel with the sizes of the row's descriptor)
__global__ __launch_bounds__(256) void pusch_cqi_encode_grants_kernel(const uint8_t* __restrict__ cqi, uint8_t* __restrict__ qb, int q_stride,
                                                                      const PuschTxDesc* __restrict__ desc, const TxDesc* __restrict__ td)
{
  __shared__ uint8_t msg[72], enc[3 * 72];
  const int          p = blockIdx.x, tid = threadIdx.x, O = desc[p].cqi_O, Q = desc[p].Qp_cqi * td[p].Qm;
  if (O == 0) return;
  const uint8_t* in  = cqi + (size_t)p * 64;
  uint8_t*       out = qb + (size_t)p * q_stride;
  if (O <= 11) {
    if (tid < 32) {
      int b = 0;
      for (int n = 0; n < O; n++) b ^= in[n] & (CQI_M32[tid] >> n) & 1;
      enc[tid] = (uint8_t)b;
    }
    __syncthreads();
    for (int i = tid; i < Q; i += 256) out[i] = enc[i & 31];
    return;
  }
  const int F = O + 8;
  if (tid < O) msg[tid] = in[tid] & 1;
  __syncthreads();
  if (tid == 0) {
    const uint32_t c = cqi_crc8(msg, O);
    for (int i = 0; i < 8; i++) msg[O + i] = (c >> (7 - i)) & 1;
  }
  __syncthreads();
  for (int e = tid; e < 3 * F; e += 256) {
    const int      i = e / 3, pp = e - 3 * i;
    const uint32_t poly = pp == 0 ? 0x6Du : (pp == 1 ? 0x4Fu : 0x57u);
    int            b = 0;
    for (int j = 0; j < 7; j++) b ^= ((poly >> j) & 1u) & msg[(i - j + F) % F];
    enc[e] = (uint8_t)b;
  }
  __syncthreads();
  const uint16_t* w = desc[p].cqi_w;
  const int       wlen = desc[p].cqi_wlen;
  for (int i = tid; i < Q; i += 256) out[i] = enc[w[i % wlen]];
}

// grid = (ceil(max M_sc / 256), nsymb, nof_pusch)
__global__ __launch_bounds__(256) void pusch_tx_mod_grants_kernel(const uint8_t* __restrict__ cb, const uint8_t* __restrict__ parity,
                                                                  const uint8_t* __restrict__ sys_tail, const uint32_t* __restrict__ scr, int scr_words,
                                                                  cf32* __restrict__ d, const PuschTxDesc* __restrict__ desc, const TxDesc* __restrict__ td,
                                                                  TxLevels lv, int nsymb, int cb_stride, int par_stride, const uint8_t* __restrict__ ack,
                                                                  const uint8_t* __restrict__ rib, const uint8_t* __restrict__ qcqi, int cqi_stride)
{
  const int          p = blockIdx.z;
  const PuschTxDesc& pd = desc[p];
  PuschTxGeom        g; // a local built from the descriptors (its lvl member stays untouched: the levels come through lv)
  g.M_sc = pd.M_sc; g.Qm = td[p].Qm; g.C = td[p].C; g.rm_len = 3 * td[p].K + 12; g.syms_lo = pd.syms_lo; g.C_lo = pd.C_lo; g.nsymb = nsymb;
  g.cb_stride = cb_stride; g.par_stride = par_stride; g.ack = pd.ack; g.ri = pd.ri; g.Qp_cqi = pd.Qp_cqi;
  pusch_tx_mod_body(cb, parity, sys_tail, td[p].rm, scr + (size_t)p * scr_words, d + pd.zoff, g, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y, td[p].cb0,
                    pd.ack.O ? ack + 2 * p : nullptr, pd.ri.O ? rib + 2 * p : nullptr, pd.cqi_O ? qcqi + (size_t)p * cqi_stride : nullptr, lv.v[td[p].mod]);
}

// grid = (ceil(max M_sc / 256), 14, nof_pusch): pusch_put + srslte_refsignal_dmrs_pusch_put of PUSCH p into the (zeroed) grid of its subframe
__global__ __launch_bounds__(256) void pusch_tx_scatter_kernel(const cf32* __restrict__ z, cf32* __restrict__ grid, const PuschTxDesc* __restrict__ desc,
                                                               const TxDesc* __restrict__ td, int cell_nre, int nsymb)
{
  const int          kk = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y, p = blockIdx.z;
  const PuschTxDesc& pd = desc[p];
  if (kk >= pd.M_sc) return;
  cf32 v;
  if (l == 3 || l == 10) {
    v = pd.dmrs[((size_t)pd.sf_idx * 2 + (l == 10)) * pd.M_sc + kk];
  } else {
    const int n = l < 3 ? l : (l < 10 ? l - 1 : l - 2);
    if (n >= nsymb) return; // the last symbol of a shortened subframe stays empty
    v = z[(size_t)pd.zoff + (size_t)n * pd.M_sc + kk];
  }
  grid[((size_t)td[p].sf * 14 + l) * cell_nre + 12 * (l < 7 ? pd.n_prb : pd.n_prb1) + kk] = v;
}

} // namespace

struct UlTxGrantsState {
  uint32_t  V, Cmax, words, cb_stride, par_stride, max_sym, cqi_stride;
  uint32_t *d_scr, *d_basis, *d_tbcrc;
  uint8_t * d_cb, *d_parity, *d_sys_tail, *d_desc, *d_qcqi;
  cf32 *    d_d, *d_z;
  size_t    desc_bytes;
  uint8_t*   h_pin[4];
  hipEvent_t h_ev[4];
  bool       h_used[4];
  uint32_t   h_slot;
  TxLevels   lv;
  std::map<std::pair<uint32_t, uint32_t>, uint32_t*> rm_tbl; // (K, rv)
  std::map<uint32_t, std::pair<uint16_t*, uint32_t>> cqi_w;  // report size O > 11 -> (device table, length)
};

static void ul_tx_grants_free(UlTxGrantsState* g)
{
  if (!g) return;
  void* gb[] = {g->d_scr, g->d_basis, g->d_tbcrc, g->d_cb, g->d_parity, g->d_sys_tail, g->d_desc, g->d_qcqi, g->d_d, g->d_z};
  for (void* b : gb) {
    if (b) (void)hipFree(b);
  }
  for (auto& kv : g->rm_tbl) (void)hipFree(kv.second);
  for (auto& kv : g->cqi_w) (void)hipFree(kv.second.first);
  for (int i = 0; i < 4; i++) {
    if (g->h_pin[i]) {
      (void)hipHostFree(g->h_pin[i]);
      (void)hipEventDestroy(g->h_ev[i]);
    }
  }
  delete g;
}

static int ul_tx_grants_init(srslte_hip_ul_tx_t* q, uint32_t V)
{
  const uint32_t P = q->cfg.nof_prb, nsymb = (uint32_t)q->g.nsymb;
  auto*          g = new UlTxGrantsState(); // value-initialised
  q->gs         = g;
  g->V          = V;
  g->Cmax       = q->seg.C;
  g->max_sym    = nsymb * 12 * P;
  g->words      = (g->max_sym * 6 + 31) / 32 + 2;
  g->cb_stride  = (6144 / 8 + 15) & ~15u;
  g->par_stride = (6144 / 4 + 1 + 15) & ~15u;
  g->cqi_stride = (g->max_sym * 6 + 15) & ~15u; // a report may take the whole allocation (min(.., M_sc N_symb - Q'_ri), uci.c:264-281)
  const size_t nblk = (size_t)V * g->Cmax;
  g->desc_bytes     = (sizeof(GrantDev) + sizeof(TxDesc) + sizeof(PuschTxDesc)) * V;
  for (int i = 0; i < 4; i++) {
    HIP_TRY(hipEventCreateWithFlags(&g->h_ev[i], hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void**)&g->h_pin[i], g->desc_bytes));
  }
  if (gold_basis_upload(g->words, &g->d_basis)) return SRSLTE_ERROR;
  HIP_TRY(hipMalloc((void**)&g->d_scr, sizeof(uint32_t) * (size_t)g->words * V));
  HIP_TRY(hipMalloc((void**)&g->d_tbcrc, sizeof(uint32_t) * V));
  HIP_TRY(hipMalloc((void**)&g->d_cb, (size_t)g->cb_stride * nblk));
  HIP_TRY(hipMalloc((void**)&g->d_parity, (size_t)g->par_stride * nblk));
  HIP_TRY(hipMalloc((void**)&g->d_sys_tail, nblk));
  HIP_TRY(hipMalloc((void**)&g->d_desc, g->desc_bytes));
  HIP_TRY(hipMalloc((void**)&g->d_qcqi, (size_t)g->cqi_stride * V));
  HIP_TRY(hipMalloc((void**)&g->d_d, sizeof(cf32) * (size_t)g->max_sym * V));
  HIP_TRY(hipMalloc((void**)&g->d_z, sizeof(cf32) * (size_t)g->max_sym * V));
  for (int mod = 1; mod <= 4; mod++) { // 36.211 7.1.2-7.1.5, one axis (lte_tables.c:57-262)
    for (uint32_t idx = 0; idx < (1u << mod); idx++) {
      double v = 1.0;
      for (int i = mod - 1; i >= 1; i--) v = (double)(1 << (mod - i)) - (1 - 2 * (int)((idx >> (mod - 1 - i)) & 1)) * v;
      const double norm = mod == 1 ? sqrt(2.0) : (mod == 2 ? sqrt(10.0) : (mod == 3 ? sqrt(42.0) : sqrt(170.0)));
      g->lv.v[mod][idx] = (float)((1 - 2 * (int)((idx >> (mod - 1)) & 1)) * v / norm);
    }
  }
  return SRSLTE_SUCCESS;
}

// grants[p]: as srslte_hip_ul_rx_batch_grants takes them (new_data is not used). Row p of d_tb is its transport block; d_ack / d_ri: [nof_grants][2],
// d_cqi: [nof_grants][64] device bytes, rows p (each may be NULL when no grant of the call carries that kind of UCI). The object's cell, DMRS
// configuration and shortened flag apply; cfg.tbs bounds every grant's tbs, cfg.max_grants the number of PUSCHs per call.
extern "C" int srslte_hip_ul_tx_batch_grants(srslte_hip_ul_tx_t* q, const uint8_t* d_tb, uint32_t tb_stride, const uint8_t* d_ack, const uint8_t* d_ri,
                                             const uint8_t* d_cqi, uint32_t tti0, uint32_t nof_sf, const srslte_hip_ul_grant_t* grants, uint32_t nof_grants,
                                             void* d_iq, void* stream)
{
  if (!q || !d_tb || !d_iq || !grants || nof_sf > q->cfg.max_batch) return SRSLTE_ERROR_INVALID_INPUTS;
  const uint32_t V = q->cfg.max_grants ? q->cfg.max_grants : q->cfg.max_batch, P = q->cfg.nof_prb, nsymb = (uint32_t)q->g.nsymb;
  if (nof_grants > V) return SRSLTE_ERROR_INVALID_INPUTS;
  if (nof_sf == 0) return SRSLTE_SUCCESS;
  hipStream_t st = (hipStream_t)stream;
  if (!q->gs && ul_tx_grants_init(q, V)) { // a failed start leaves no half-made state behind
    ul_tx_grants_free(q->gs);
    q->gs = nullptr;
    return SRSLTE_ERROR;
  }
  UlTxGrantsState* g  = q->gs;
  const uint32_t   hs = g->h_slot++ & 3u;
  if (g->h_used[hs]) HIP_TRY(hipEventSynchronize(g->h_ev[hs]));
  auto* h_gr = reinterpret_cast<GrantDev*>(g->h_pin[hs]);
  auto* h_td = reinterpret_cast<TxDesc*>(h_gr + V);
  auto* h_pd = reinterpret_cast<PuschTxDesc*>(h_td + V);
  auto* d_gr = reinterpret_cast<GrantDev*>(g->d_desc);
  auto* d_td = reinterpret_cast<TxDesc*>(d_gr + V);
  auto* d_pd = reinterpret_cast<PuschTxDesc*>(d_td + V);
  std::vector<srslte_hip_cbsegm_t> segs(nof_grants);
  std::vector<uint32_t>            by_k(nof_grants), by_l(nof_grants);
  uint32_t                         max_M = 0;
  bool                             any_cqi = false;
  for (uint32_t p = 0; p < nof_grants; p++) {
    const srslte_hip_ul_grant_t& gr = grants[p];
    by_k[p] = by_l[p] = p;
    if (gr.sf >= nof_sf || gr.L_prb == 0 || !srslte_hip_dft_precoding_valid_prb(gr.L_prb) || gr.n_prb + gr.L_prb > P || gr.n_prb_slot1 + gr.L_prb > P ||
        gr.n_dmrs >= 8 || gr.mod < 1 || gr.mod > 3 || gr.rv > 3 || gr.tbs == 0 || gr.tbs > q->cfg.tbs || (gr.tbs % 8) || tb_stride < gr.tbs / 8 ||
        srslte_hip_cbsegm(&segs[p], gr.tbs) || segs[p].F || segs[p].C2 || segs[p].C > g->Cmax || (gr.ack_len && !d_ack) || (gr.ri_len && !d_ri) ||
        (gr.cqi_len && !d_cqi)) {
      hip_log("[srslte_hip] ul_tx grants: entry %u: unsupported grant (subframe %u of %u, L_prb %u at %u / %u, mod %d, tbs %u, rv %u)\n", p, gr.sf, nof_sf, gr.L_prb,
              gr.n_prb, gr.n_prb_slot1, gr.mod, gr.tbs, gr.rv);
      return SRSLTE_ERROR_INVALID_INPUTS;
    }
    max_M = 12 * gr.L_prb > max_M ? 12 * gr.L_prb : max_M;
    any_cqi = any_cqi || gr.cqi_len;
  }
  // code-block slots in block-length order (one encoder launch per length), symbol buffers in L_prb order (one transform-precoding launch per size)
  std::stable_sort(by_k.begin(), by_k.end(), [&](uint32_t a, uint32_t b) { return segs[a].K1 < segs[b].K1; });
  std::stable_sort(by_l.begin(), by_l.end(), [&](uint32_t a, uint32_t b) { return grants[a].L_prb < grants[b].L_prb; });
  uint32_t cb0 = 0, zoff = 0;
  for (uint32_t i = 0; i < nof_grants; i++) {
    h_td[by_k[i]].cb0 = (int)cb0;
    cb0 += segs[by_k[i]].C;
    h_pd[by_l[i]].zoff = (int)zoff;
    zoff += nsymb * 12 * grants[by_l[i]].L_prb;
  }
  for (uint32_t p = 0; p < nof_grants; p++) {
    const srslte_hip_ul_grant_t& gr = grants[p];
    const uint32_t               K = segs[p].K1, C = segs[p].C, nof_re = nsymb * 12 * gr.L_prb;
    GrantDev&                    gd = h_gr[p];
    memset(&gd, 0, sizeof(gd));
    gd.sf_idx = (int)((tti0 + gr.sf) % 10); gd.rnti = gr.rnti;
    const int Qp_ack = pusch_ack_qprime(gr.ack_len, gr.I_offset_ack, gr.L_prb, nsymb, C * K);
    const int Qp_ri  = pusch_ack_qprime(gr.ri_len, gr.I_offset_ri, gr.L_prb, nsymb, C * K, true);
    const int Qp_cqi = Qp_ri >= 0 && gr.cqi_len <= 64 ? pusch_cqi_qprime(gr.cqi_len, gr.I_offset_cqi, gr.L_prb, nsymb, C * K, (uint32_t)Qp_ri) : -1;
    if (Qp_ack < 0 || Qp_ri < 0 || Qp_cqi < 0 || (uint32_t)(Qp_ri + Qp_cqi) + C >= nof_re) {
      hip_log("[srslte_hip] ul_tx grants: entry %u: invalid UCI configuration\n", p);
      return SRSLTE_ERROR_INVALID_INPUTS;
    }
    auto it = g->rm_tbl.find({K, gr.rv});
    if (it == g->rm_tbl.end()) {
      std::vector<uint32_t> t;
      lte_rm_rx_table(K, gr.rv, t);
      for (auto& v : t) {
        const uint32_t pos = v / 3, sidx = v % 3;
        v = sidx == 0 ? (pos < K ? pos : (1u << 30) | (pos - K)) : (2u << 30) | (sidx == 1 ? pos : K + 4 + pos);
      }
      uint32_t* d = nullptr;
      if (upload(&d, t)) return SRSLTE_ERROR;
      it = g->rm_tbl.emplace(std::make_pair(K, gr.rv), d).first;
    }
    TxDesc& td = h_td[p];
    td.row = (int)p; td.sf = (int)gr.sf; td.tbs = (int)gr.tbs; td.C = (int)C; td.K = (int)K; td.rlenB = (int)((C == 1 ? K : K - 24) / 8);
    td.nre = (int)nof_re; td.mod = gr.mod; td.Qm = 2 * gr.mod; td.rm = it->second;
    PuschTxDesc& pd = h_pd[p];
    const uint32_t g_re = nof_re - (uint32_t)Qp_ri - (uint32_t)Qp_cqi; // UL-SCH symbols (sch.c:1157-1160)
    pd.M_sc = 12 * (int)gr.L_prb; pd.n_prb = (int)gr.n_prb; pd.n_prb1 = (int)gr.n_prb_slot1; pd.syms_lo = (int)(g_re / C); pd.C_lo = (int)(C - g_re % C);
    pd.Qp_cqi = Qp_cqi; pd.cqi_O = (int)gr.cqi_len; pd.cqi_w = nullptr; pd.cqi_wlen = 1; pd.sf_idx = gd.sf_idx;
    pd.ack.O = (int)gr.ack_len; pd.ack.Qprime = Qp_ack; pd.ri.O = (int)gr.ri_len; pd.ri.Qprime = Qp_ri;
    if (gr.cqi_len > 11) { // srslte_rm_conv_tx (rm_conv.c:44-89): the sub-block interleaved streams without their dummies, read circularly
      auto cw = g->cqi_w.find(gr.cqi_len);
      if (cw == g->cqi_w.end()) {
        static const uint8_t perm[32] = {1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31, 0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30};
        const int             F = (int)gr.cqi_len + 8, nrows = (F - 1) / 32 + 1, ndummy = nrows * 32 - F;
        std::vector<uint16_t> w;
        for (int s3 = 0; s3 < 3; s3++) {
          for (int j = 0; j < 32; j++) {
            for (int i = 0; i < nrows; i++) {
              const int pos = i * 32 + perm[j];
              if (pos >= ndummy) w.push_back((uint16_t)((pos - ndummy) * 3 + s3));
            }
          }
        }
        uint16_t* d = nullptr;
        if (upload(&d, w)) return SRSLTE_ERROR;
        cw = g->cqi_w.emplace(gr.cqi_len, std::make_pair(d, (uint32_t)w.size())).first;
      }
      pd.cqi_w = cw->second.first; pd.cqi_wlen = (int)cw->second.second;
    }
    const void* d_r = nullptr;
    if (int r = chest_ul_dmrs_table_cached(q->dmrs, gr.L_prb, gr.n_dmrs, &d_r)) return r;
    pd.dmrs = (const cf32*)d_r;
  }
  HIP_TRY(hipMemcpyAsync(g->d_desc, g->h_pin[hs], g->desc_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(g->h_ev[hs], st));
  g->h_used[hs] = true;
  HIP_TRY(hipMemsetAsync(q->d_grid, 0, sizeof(cf32) * (size_t)14 * 12 * P * nof_sf, st)); // ue_ul.c:320: the grid is cleared, then pusch_put
  if (nof_grants) {
    hipLaunchKernelGGL(scr_gen_kernel, dim3(ceil_div((int)g->words, 256), nof_grants), dim3(256), 0, st, (const GrantDev*)d_gr, (const uint32_t*)g->d_basis,
                       g->d_scr, (int)g->words, (int)q->cfg.cell_id);
    if (any_cqi) {
      hipLaunchKernelGGL(pusch_cqi_encode_grants_kernel, dim3(nof_grants), dim3(256), 0, st, d_cqi, g->d_qcqi, (int)g->cqi_stride, (const PuschTxDesc*)d_pd,
                         (const TxDesc*)d_td);
    }
    hipLaunchKernelGGL(tx_tbcrc_grants_kernel, dim3(nof_grants), dim3(256), 0, st, d_tb, (int)tb_stride, (const TxDesc*)d_td, g->d_tbcrc);
    hipLaunchKernelGGL(tx_seg_grants_kernel, dim3(g->Cmax, nof_grants), dim3(256), 0, st, d_tb, (int)tb_stride, (const uint32_t*)g->d_tbcrc, (const TxDesc*)d_td,
                       g->d_cb, (int)g->cb_stride);
    LAUNCH_CHECK();
    for (uint32_t i = 0; i < nof_grants;) { // the encoder: runs of equal block length
      uint32_t j = i, n = 0;
      while (j < nof_grants && segs[by_k[j]].K1 == segs[by_k[i]].K1) n += segs[by_k[j++]].C;
      const size_t s0 = (size_t)h_td[by_k[i]].cb0;
      if (int r = srslte_hip_tcod_encode_bytes_batch(g->d_cb + s0 * g->cb_stride, g->cb_stride, g->d_parity + s0 * g->par_stride, g->par_stride,
                                                     g->d_sys_tail + s0, segs[by_k[i]].K1, n, stream))
        return r;
      i = j;
    }
    hipLaunchKernelGGL(pusch_tx_mod_grants_kernel, dim3(ceil_div((int)max_M, 256), nsymb, nof_grants), dim3(256), 0, st, (const uint8_t*)g->d_cb,
                       (const uint8_t*)g->d_parity, (const uint8_t*)g->d_sys_tail, (const uint32_t*)g->d_scr, (int)g->words, g->d_d, (const PuschTxDesc*)d_pd,
                       (const TxDesc*)d_td, g->lv, (int)nsymb, (int)g->cb_stride, (int)g->par_stride, d_ack, d_ri, (const uint8_t*)g->d_qcqi, (int)g->cqi_stride);
    LAUNCH_CHECK();
    for (uint32_t i = 0; i < nof_grants;) { // transform precoding: runs of equal L_prb (srslte_dft_precoding_init_tx: forward, 1/sqrt(N))
      uint32_t j = i + 1;
      while (j < nof_grants && grants[by_l[j]].L_prb == grants[by_l[i]].L_prb) j++;
      const size_t off = (size_t)h_pd[by_l[i]].zoff;
      if (int r = srslte_hip_dft_precoding_batch(g->d_d + off, g->d_z + off, grants[by_l[i]].L_prb, nsymb * (j - i), 1, stream)) return r;
      i = j;
    }
    hipLaunchKernelGGL(pusch_tx_scatter_kernel, dim3(ceil_div((int)max_M, 256), 14, nof_grants), dim3(256), 0, st, (const cf32*)g->d_z, q->d_grid,
                       (const PuschTxDesc*)d_pd, (const TxDesc*)d_td, 12 * (int)P, (int)nsymb);
    LAUNCH_CHECK();
  }
  return srslte_hip_ofdm_tx_sf_batch(q->ofdm, q->d_grid, d_iq, (int)nof_sf, stream);
}
