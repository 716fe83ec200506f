// The rest of the public prototypes of ch_estimation/refsignal_dl.h and ch_estimation/chest_common.h (SURVEY §8b lists both headers in
// the boundary), plus the 25.212 interleaver generator row a6 cites, so that refsignal_dl.c, chest_common.c and tc_interl_umts.c need not
// stay in a build that links this library. What they are:
//   * index rules (which symbol / subcarrier / how many CRS symbols) and init-time tables (CRS and MBSFN-RS values from the Gold
//     sequence, filter taps): host integer / scalar code, like srslte_cbsegm and the QPP tables - nothing a device would speed up;
//   * put / get of reference symbols in a caller's HOST grid: a strided copy of <= 800 values between two host arrays;
//   * srslte_chest_average_pilots / srslte_chest_estimate_noise_pilots: array arithmetic -> the device, with the copy in / launch /
//     copy out of the other single-call wrappers (compat.cpp); no host arithmetic path exists for them.
// The batched pipeline uses none of this: its estimator kernels hold their own device tables (chest.hip).
#include "phy_hip_internal.hpp"
#include "srslte_hip/srslte_compat.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ERROR(fmt, ...) hip_log("[srslte_hip] " fmt "\n", ##__VA_ARGS__)

namespace {

inline int cp_nsymb(srslte_cp_t cp) { return cp == SRSLTE_CP_NORM ? 7 : 6; }
inline size_t re_idx(uint32_t nof_prb, uint32_t symbol, uint32_t k) { return (size_t)symbol * 12 * nof_prb + k; } // SRSLTE_RE_IDX
inline cf_t   qpsk(uint8_t b0, uint8_t b1)
{ // (1 - 2 c(2m)) / sqrt(2) + j (1 - 2 c(2m+1)) / sqrt(2), the division in double as upstream (refsignal_dl.c:103-104)
  cf_t v;
  ((float*)&v)[0] = (float)((1 - 2 * (float)b0) / sqrt(2.0));
  ((float*)&v)[1] = (float)((1 - 2 * (float)b1) / sqrt(2.0));
  return v;
}
bool cell_ok(const srslte_cell_t& c) { return c.id < 504 && c.nof_ports <= SRSLTE_MAX_PORTS && c.nof_prb >= 6 && c.nof_prb <= 100; } // phy_common.c:40-60

// 36.211 Table 4.2-2 (which subframes of a TDD frame are downlink) and the DwPTS length in symbols of Table 4.2-1 per special-subframe
// configuration (phy_common.c:91-101)
const char     TDD_KIND[7][11] = {"DSUUUDSUUU", "DSUUDDSUUD", "DSUDDDSUDD", "DSUUUDDDDD", "DSUUDDDDDD", "DSUDDDDDDD", "DSUUUDSUUD"};
const uint32_t TDD_DWPTS[10]   = {3, 9, 10, 11, 12, 3, 9, 10, 11, 6};

bool is_full_dl_subframe(const srslte_refsignal_t* q, const srslte_dl_sf_cfg_t* sf)
{ // refsignal_dl.c:164-166
  if (!q || !sf || q->cell.frame_type == SRSLTE_FDD || !sf->tdd_config.configured) return true;
  const uint32_t idx = sf->tti % 10;
  return sf->tdd_config.sf_config >= 7 || TDD_KIND[sf->tdd_config.sf_config][idx] == 'D';
}

void free_tables(srslte_refsignal_t* q)
{
  for (auto& grp : q->pilots) {
    for (auto& p : grp) {
      free(p);
      p = nullptr;
    }
  }
}

int alloc_tables(srslte_refsignal_t* q, size_t n)
{
  for (auto& grp : q->pilots) {
    for (auto& p : grp) {
      void* m = nullptr;
      if (posix_memalign(&m, 64, sizeof(cf_t) * (n ? n : 1))) {
        free_tables(q);
        return SRSLTE_ERROR;
      }
      p = (cf_t*)m;
    }
  }
  return SRSLTE_SUCCESS;
}

// per-thread staging for the two device helpers (no object to hang it on, as for the demapper)
struct Stage {
  void*  p = nullptr;
  size_t n = 0;
  void*  get(size_t bytes)
  {
    if (bytes > n) {
      if (p) (void)hipFree(p);
      p = nullptr;
      n = 0;
      if (hipMalloc(&p, bytes) != hipSuccess) {
        ERROR("hipMalloc(%zu) failed", bytes);
        return nullptr;
      }
      n = bytes;
    }
    return p;
  }
};
thread_local Stage g_a, g_b, g_c, g_f;

} // namespace

extern "C" {

// ====================================================================================================== cell-specific reference signal
uint32_t srslte_refsignal_cs_v(uint32_t port_id, uint32_t ref_symbol_idx)
{ // 36.211 6.10.1.2, refsignal_dl.c:127-160: ports 0/1 alternate with the reference symbol, ports 2/3 switch after the first
  if (port_id > 3) return 0;
  const bool second = port_id < 2 ? (ref_symbol_idx % 2) != 0 : ref_symbol_idx != 0;
  return (second != ((port_id & 1) != 0)) ? 3 : 0;
}

uint32_t srslte_refsignal_cs_nof_symbols(srslte_refsignal_t* q, srslte_dl_sf_cfg_t* sf, uint32_t port_id)
{ // refsignal_dl.c:162-225: 4 / 2 CRS symbols per subframe, fewer in the DwPTS of a TDD special (or uplink) subframe
  const bool low = port_id < 2;
  if (is_full_dl_subframe(q, sf)) return low ? 4 : 2;
  const uint32_t dw   = sf->tdd_config.ss_config < 10 ? TDD_DWPTS[sf->tdd_config.ss_config] : 0;
  const bool     norm = q->cell.cp == SRSLTE_CP_NORM;
  if (dw >= (norm ? 12u : 10u)) return low ? 4 : 2;
  if (dw >= (norm ? 9u : 8u)) return low ? 3 : 2;
  if (dw >= (norm ? 5u : 4u)) return low ? 2 : 1;
  return 1;
}

uint32_t srslte_refsignal_cs_nof_re(srslte_refsignal_t* q, srslte_dl_sf_cfg_t* sf, uint32_t port_id)
{
  return srslte_refsignal_cs_nof_symbols(q, sf, port_id) * q->cell.nof_prb * 2;
}

uint32_t srslte_refsignal_cs_fidx(srslte_cell_t cell, uint32_t l, uint32_t port_id, uint32_t m)
{
  return 6 * m + (srslte_refsignal_cs_v(port_id, l) + cell.id % 6) % 6;
}

uint32_t srslte_refsignal_cs_nsymbol(uint32_t l, srslte_cp_t cp, uint32_t port_id)
{ // refsignal_dl.c:236-247: symbols 0 and N_symb - 3 of each slot (ports 0/1), symbol 1 of each slot (ports 2/3)
  const uint32_t n = (uint32_t)cp_nsymb(cp);
  if (port_id >= 2) return 1 + l * n;
  return (l % 2) ? (l / 2 + 1) * n - 3 : (l / 2) * n;
}

int srslte_refsignal_cs_init(srslte_refsignal_t* q, uint32_t max_prb)
{ // refsignal_dl.c:37-61
  if (!q) return SRSLTE_ERROR_INVALID_INPUTS;
  memset(q, 0, sizeof(*q));
  return alloc_tables(q, (size_t)8 * max_prb); // SRSLTE_REFSIGNAL_MAX_NUM_SF
}

void srslte_refsignal_free(srslte_refsignal_t* q)
{ // refsignal_dl.c:115-125
  if (!q) return;
  free_tables(q);
  memset(q, 0, sizeof(*q));
}

int srslte_refsignal_cs_set_cell(srslte_refsignal_t* q, srslte_cell_t cell)
{ // refsignal_dl.c:66-112: r_{l,ns}(m) of 36.211 6.10.1.1 for the cell's middle 2 nof_prb values of each CRS symbol
  if (!q || !cell_ok(cell)) return SRSLTE_ERROR_INVALID_INPUTS;
  if (cell.id == q->cell.id && q->cell.nof_prb != 0) return SRSLTE_SUCCESS; // only a new cell id rebuilds, as upstream (:77)
  q->cell                 = cell;
  const uint32_t       ncp = cell.cp == SRSLTE_CP_NORM ? 1 : 0, nref = 2 * cell.nof_prb;
  std::vector<uint8_t> c;
  for (uint32_t ns = 0; ns < 20; ns++) {
    for (uint32_t grp = 0; grp < 2; grp++) {
      if (!q->pilots[grp][ns / 2]) return SRSLTE_ERROR;
      const uint32_t per_slot = grp == 0 ? 2 : 1;
      for (uint32_t l = 0; l < per_slot; l++) {
        const uint32_t lp     = srslte_refsignal_cs_nsymbol(l, cell.cp, 2 * grp);
        const uint32_t c_init = 1024 * (7 * (ns + 1) + lp + 1) * (2 * cell.id + 1) + 2 * cell.id + ncp;
        lte_gold_sequence(c_init, 4 * SRSLTE_MAX_PRB, c);
        cf_t* dst = q->pilots[grp][ns / 2] + (size_t)nref * ((ns % 2) * per_slot + l);
        for (uint32_t i = 0; i < nref; i++) {
          const uint32_t mp = i + SRSLTE_MAX_PRB - cell.nof_prb;
          dst[i]            = qpsk(c[2 * mp], c[2 * mp + 1]);
        }
      }
    }
  }
  return SRSLTE_SUCCESS;
}

int srslte_refsignal_cs_put_sf(srslte_refsignal_t* q, srslte_dl_sf_cfg_t* sf, uint32_t port_id, cf_t* sf_symbols)
{ // refsignal_dl.c:249-270
  if (!q || port_id >= SRSLTE_MAX_PORTS || !sf_symbols || !sf) return SRSLTE_ERROR_INVALID_INPUTS;
  const cf_t*    pil  = q->pilots[port_id / 2][sf->tti % 10];
  const uint32_t nref = 2 * q->cell.nof_prb, nl = srslte_refsignal_cs_nof_symbols(q, sf, port_id);
  for (uint32_t l = 0; l < nl; l++) {
    cf_t* row = sf_symbols + re_idx(q->cell.nof_prb, srslte_refsignal_cs_nsymbol(l, q->cell.cp, port_id), srslte_refsignal_cs_fidx(q->cell, l, port_id, 0));
    for (uint32_t i = 0; i < nref; i++) row[6 * i] = pil[(size_t)nref * l + i];
  }
  return SRSLTE_SUCCESS;
}

int srslte_refsignal_cs_get_sf(srslte_refsignal_t* q, srslte_dl_sf_cfg_t* sf, uint32_t port_id, cf_t* sf_symbols, cf_t* pilots)
{ // refsignal_dl.c:273-293
  if (!q || !pilots || !sf_symbols) return SRSLTE_ERROR_INVALID_INPUTS;
  const uint32_t nref = 2 * q->cell.nof_prb, nl = srslte_refsignal_cs_nof_symbols(q, sf, port_id);
  for (uint32_t l = 0; l < nl; l++) {
    const cf_t* row = sf_symbols + re_idx(q->cell.nof_prb, srslte_refsignal_cs_nsymbol(l, q->cell.cp, port_id), srslte_refsignal_cs_fidx(q->cell, l, port_id, 0));
    for (uint32_t i = 0; i < nref; i++) pilots[(size_t)nref * l + i] = row[6 * i];
  }
  return SRSLTE_SUCCESS;
}

// ====================================================================================================== MBSFN reference signal (port 4)
uint32_t srslte_refsignal_mbsfn_nof_symbols() { return 3; }
uint32_t srslte_refsignal_mbsfn_fidx(uint32_t l) { return l == 1 ? 1 : 0; }                    // refsignal_dl.c:327-340: k = 2m + 1 in symbol 6
uint32_t srslte_refsignal_mbsfn_nsymbol(uint32_t l) { return l == 0 ? 2 : (l == 1 ? 6 : (l == 2 ? 10 : 0)); } // extended CP: symbols 2, 6, 10

int srslte_refsignal_mbsfn_init(srslte_refsignal_t* q, uint32_t max_prb)
{ // refsignal_dl.c:403-431
  if (!q) return SRSLTE_ERROR_INVALID_INPUTS;
  memset(q, 0, sizeof(*q));
  q->type = SRSLTE_SF_MBSFN;
  return alloc_tables(q, (size_t)18 * max_prb);
}

int srslte_refsignal_mbsfn_gen_seq(srslte_refsignal_t* q, srslte_cell_t cell, uint32_t N_mbsfn_id)
{ // refsignal_dl.c:361-400, 36.211 6.10.2.1: the table is indexed with q's OWN cell width, the sequence offset with the argument's (:384-385)
  if (!q) return SRSLTE_ERROR;
  const uint32_t       nmb = 6 * q->cell.nof_prb;
  std::vector<uint8_t> c;
  for (uint32_t sf = 0; sf < 10; sf++) {
    for (uint32_t l = 0; l < 3; l++) {
      const uint32_t lp = srslte_refsignal_mbsfn_nsymbol(l) % 6, slot = l ? 2 * sf + 1 : 2 * sf;
      const uint32_t c_init = 512 * (7 * (slot + 1) + lp + 1) * (2 * N_mbsfn_id + 1) + N_mbsfn_id;
      lte_gold_sequence(c_init, 20 * SRSLTE_MAX_PRB, c);
      for (uint32_t grp = 0; grp < 2; grp++) {
        if (!q->pilots[grp][sf]) return SRSLTE_ERROR;
        for (uint32_t i = 0; i < nmb; i++) {
          const uint32_t mp                       = i + 3 * (SRSLTE_MAX_PRB - cell.nof_prb);
          q->pilots[grp][sf][(size_t)nmb * l + i] = qpsk(c[2 * mp], c[2 * mp + 1]);
        }
      }
    }
  }
  return SRSLTE_SUCCESS;
}

int srslte_refsignal_mbsfn_set_cell(srslte_refsignal_t* q, srslte_cell_t cell, uint16_t mbsfn_area_id)
{ // refsignal_dl.c:433-452
  if (!q) return SRSLTE_ERROR_INVALID_INPUTS;
  q->cell          = cell;
  q->mbsfn_area_id = mbsfn_area_id;
  if (srslte_refsignal_mbsfn_gen_seq(q, q->cell, q->mbsfn_area_id)) {
    srslte_refsignal_free(q);
    return SRSLTE_ERROR;
  }
  return SRSLTE_SUCCESS;
}

int srslte_refsignal_mbsfn_get_sf(srslte_cell_t cell, uint32_t port_id, cf_t* sf_symbols, cf_t* pilots)
{ // refsignal_dl.c:455-487: the CRS of symbol 0 (non-MBSFN region) first, the three MBSFN rows behind them
  if (!cell_ok(cell) || port_id > SRSLTE_MAX_PORTS || !pilots || !sf_symbols) return SRSLTE_ERROR_INVALID_INPUTS;
  const uint32_t nref = 2 * cell.nof_prb, nmb = 6 * cell.nof_prb;
  const cf_t*    row  = sf_symbols + re_idx(cell.nof_prb, srslte_refsignal_cs_nsymbol(0, cell.cp, port_id), (srslte_refsignal_cs_v(port_id, 0) + cell.id % 6) % 6);
  for (uint32_t i = 0; i < nref; i++) pilots[i] = row[6 * i];
  for (uint32_t l = 0; l < 3; l++) {
    row = sf_symbols + re_idx(cell.nof_prb, srslte_refsignal_mbsfn_nsymbol(l), srslte_refsignal_mbsfn_fidx(l));
    for (uint32_t i = 0; i < nmb; i++) pilots[nref + (size_t)nmb * l + i] = row[2 * i];
  }
  return SRSLTE_SUCCESS;
}

int srslte_refsignal_mbsfn_put_sf(srslte_cell_t cell, uint32_t port_id, cf_t* cs_pilots, cf_t* mbsfn_pilots, cf_t* sf_symbols)
{ // refsignal_dl.c:296-325: symbol 0 takes the CRS, symbols 2 / 6 / 10 the MBSFN reference
  if (!cell_ok(cell) || port_id > SRSLTE_MAX_PORTS || !cs_pilots || !mbsfn_pilots || !sf_symbols) return SRSLTE_ERROR_INVALID_INPUTS;
  const uint32_t nref = 2 * cell.nof_prb, nmb = 6 * cell.nof_prb;
  cf_t*          row  = sf_symbols + re_idx(cell.nof_prb, 0, (srslte_refsignal_cs_v(port_id, 0) + cell.id % 6) % 6);
  for (uint32_t i = 0; i < nref; i++) row[6 * i] = cs_pilots[i];
  for (uint32_t l = 0; l < 3; l++) {
    row = sf_symbols + re_idx(cell.nof_prb, srslte_refsignal_mbsfn_nsymbol(l), srslte_refsignal_mbsfn_fidx(l));
    for (uint32_t i = 0; i < nmb; i++) row[2 * i] = mbsfn_pilots[(size_t)nmb * l + i];
  }
  return SRSLTE_SUCCESS;
}

// ====================================================================================================== chest_common.h
uint32_t srslte_chest_set_triangle_filter(float* fil, int filter_len)
{ // chest_common.c:32-48: 1, 2, .. n/2+1 .. 2, 1 over their sum
  const int h = filter_len / 2;
  for (int i = 0; i < h; i++) {
    fil[i] = (float)(i + 1);
    if (i + h + 1 < filter_len) fil[i + h + 1] = (float)(h - i); // an even length makes upstream write one float past the array; not here
  }
  fil[h]  = (float)(h + 1);
  float s = 0;
  for (int i = 0; i < filter_len; i++) s += fil[i];
  for (int i = 0; i < filter_len; i++) fil[i] /= s;
  return (uint32_t)filter_len;
}

uint32_t srslte_chest_set_smooth_filter3_coeff(float* smooth_filter, float w)
{ // chest_common.c:62-68
  smooth_filter[0] = w;
  smooth_filter[2] = w;
  smooth_filter[1] = 1 - 2 * w;
  return 3;
}

uint32_t srslte_chest_set_smooth_filter_gauss(float* filter, uint32_t order, float std_dev)
{ // chest_common.c:70-88: exp(-(i - c)^2 / (2 sigma^2)), unit sum
  const uint32_t len = order + 1;
  if (!len) return 0;
  const int center = (int)(len - 1) / 2;
  float     sum    = 0;
  for (uint32_t i = 0; i < len; i++) {
    filter[i] = expf(-powf((float)((int)i - center), 2) / (2.0f * powf(std_dev, 2)));
    sum += filter[i];
  }
  for (uint32_t i = 0; i < len; i++) filter[i] *= 1.0f / sum;
  return len;
}

void srslte_chest_average_pilots(cf_t* input, cf_t* output, float* filter, uint32_t nof_ref, uint32_t nof_symbols, uint32_t filter_len)
{ // chest_common.c:95-101 on the device. srslte_conv_same_cf reads input[0 .. filter_len) at either end (convolution.c:186-203)
  if (!input || !output || !filter || !nof_ref || !nof_symbols) return;
  if (filter_len == 0 || filter_len > nof_ref || nof_ref < 2) {
    ERROR("srslte_chest_average_pilots: filter of %u taps over rows of %u", filter_len, nof_ref);
    return;
  }
  const size_t n  = sizeof(cf_t) * nof_ref * nof_symbols;
  void *       di = g_a.get(n), *dout = g_b.get(n), *df = g_f.get(sizeof(float) * filter_len);
  if (!di || !dout || !df || hipMemcpy(di, input, n, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(df, filter, sizeof(float) * filter_len, hipMemcpyHostToDevice) != hipSuccess ||
      chest_average_pilots_launch(di, dout, (const float*)df, (int)nof_ref, (int)nof_symbols, (int)filter_len, nullptr) ||
      hipMemcpy(output, dout, n, hipMemcpyDeviceToHost) != hipSuccess)
    ERROR("srslte_chest_average_pilots: device call failed");
}

float srslte_chest_estimate_noise_pilots(cf_t* noisy, cf_t* noiseless, cf_t* noise_vec, uint32_t nof_pilots)
{ // chest_common.c:51-60 on the device
  if (!noisy || !noiseless || !noise_vec || !nof_pilots) return 0.f;
  const size_t n  = sizeof(cf_t) * nof_pilots;
  void *       da = g_a.get(n), *db = g_b.get(n), *dc = g_c.get(n);
  float*       dp = (float*)g_f.get(sizeof(float));
  float        power = 0.f;
  if (!da || !db || !dc || !dp || hipMemcpy(da, noisy, n, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(db, noiseless, n, hipMemcpyHostToDevice) != hipSuccess || chest_noise_pilots_launch(da, db, dc, (int)nof_pilots, dp, nullptr) ||
      hipMemcpy(noise_vec, dc, n, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&power, dp, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
    ERROR("srslte_chest_estimate_noise_pilots: device call failed");
    return 0.f;
  }
  return power;
}

// ====================================================================================================== 25.212 turbo interleaver
int srslte_tc_interl_init(srslte_tc_interl_t* h, uint32_t max_long_cb)
{ // tc_interl_umts.c-style allocation used by both interleavers
  h->forward = (uint16_t*)calloc(max_long_cb, sizeof(uint16_t));
  h->reverse = (uint16_t*)calloc(max_long_cb, sizeof(uint16_t));
  if (!h->forward || !h->reverse) {
    free(h->forward);
    free(h->reverse);
    return SRSLTE_ERROR;
  }
  h->max_long_cb = max_long_cb;
  return SRSLTE_SUCCESS;
}
void srslte_tc_interl_free(srslte_tc_interl_t* h)
{
  free(h->forward);
  free(h->reverse);
  memset(h, 0, sizeof(*h));
}
// srslte_tc_interl_UMTS_gen (tc_interl_umts.c:80-262): the 3GPP TS 25.212 4.2.3.2.3 prime interleaver as the reference builds it. Two
// things differ from the specification's text and are kept, since the reference is what a caller of this symbol gets today: the row
// multipliers q_i are the least INTEGERS > 6 coprime with p - 1 in increasing order (the text asks for primes; they differ from q = 25
// on), and output row i reads source row T(i) with the column permutation of row i (not of row T(i)).
int srslte_tc_interl_UMTS_gen(srslte_tc_interl_t* h, uint32_t long_cb)
{
  static const uint16_t PRIME[52] = {7,   11,  13,  17,  19,  23,  29,  31,  37,  41,  43,  47,  53,  59,  61,  67,  71,  73,
                                     79,  83,  89,  97,  101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167,
                                     173, 179, 181, 191, 193, 197, 199, 211, 223, 227, 229, 233, 239, 241, 251, 257}; // 25.212 Table 2
  static const uint8_t  ROOT[52]  = {3, 2, 2, 3, 2, 5, 2, 3, 2, 6, 3, 5, 2, 2, 2, 2, 7, 5, 3, 2, 3, 5, 2, 5, 2, 6,
                                     3, 3, 2, 3, 2, 2, 6, 5, 2, 5, 2, 2, 2, 19, 5, 2, 3, 2, 3, 2, 6, 3, 7, 7, 6, 3}; // primitive roots v
  // inter-row permutation patterns, 25.212 Table 3
  static const uint8_t T5[5]   = {4, 3, 2, 1, 0};
  static const uint8_t T10[10] = {9, 8, 7, 6, 5, 4, 3, 2, 1, 0};
  static const uint8_t T20A[20] = {19, 9, 14, 4, 0, 2, 5, 7, 12, 18, 16, 13, 17, 15, 3, 1, 6, 11, 8, 10};
  static const uint8_t T20B[20] = {19, 9, 14, 4, 0, 2, 5, 7, 12, 18, 10, 8, 13, 17, 3, 1, 16, 6, 15, 11};
  if (!h || !h->forward || !h->reverse) return SRSLTE_ERROR;
  if (long_cb > h->max_long_cb) {
    ERROR("Interleaver initiated for max_long_cb=%u", h->max_long_cb);
    return SRSLTE_ERROR;
  }
  const uint32_t K = long_cb;
  if (K < 40 || K > 5114) { // the block sizes 25.212 defines; upstream runs off the end of its prime table above
    ERROR("UMTS interleaver: invalid block size %u", K);
    return SRSLTE_ERROR;
  }
  const bool     mid = K >= 481 && K <= 530;
  const uint32_t R   = K <= 159 ? 5 : ((K <= 200 || mid) ? 10 : 20);
  const uint8_t* T   = R == 5 ? T5 : (R == 10 ? T10 : (((K >= 2281 && K <= 2480) || (K >= 3161 && K <= 3210)) ? T20A : T20B));
  uint32_t       p = 53, v = 2;
  if (!mid) {
    int t = 0;
    while (K > R * ((uint32_t)PRIME[t] + 1)) t++;
    p = PRIME[t];
    v = ROOT[t];
  }
  const uint32_t C = K <= R * (p - 1) ? p - 1 : (K <= R * p ? p : p + 1);

  std::vector<uint32_t> s(p - 1), r(R);
  s[0] = 1;
  for (uint32_t j = 1; j < p - 1; j++) s[j] = (v * s[j - 1]) % p;
  auto gcd = [](uint32_t a, uint32_t b) {
    while (b) {
      const uint32_t t = a % b;
      a                = b;
      b                = t;
    }
    return a;
  };
  uint32_t q = 1, next = 6;
  for (uint32_t i = 0; i < R; i++) {
    if (i) {
      do next++;
      while (gcd(next, p - 1) != 1);
      q = next;
    }
    r[T[i]] = q;
  }
  // intra-row permutation of row i, column j
  auto U = [&](uint32_t i, uint32_t j) -> uint32_t {
    if (j < p - 1) return s[(j * r[i]) % (p - 1)] - (C == p - 1 ? 1 : 0);
    if (j == p - 1) return 0; // C >= p
    return p;                 // C == p + 1, j == p
  };
  const bool swap_last = C == p + 1 && K == R * C; // 25.212: exchange U_{R-1}(p) and U_{R-1}(0)
  uint32_t   k         = 0;
  for (uint32_t j = 0; j < C; j++) {
    for (uint32_t i = 0; i < R; i++) {
      uint32_t u = U(i, j);
      if (swap_last && i == R - 1 && (j == 0 || j == p)) u = U(i, j == 0 ? p : 0);
      const uint32_t src = T[i] * C + u;
      if (src < K) {
        h->reverse[src] = (uint16_t)k;
        h->forward[k]   = (uint16_t)src;
        k++;
      }
    }
  }
  return SRSLTE_SUCCESS;
}

} // extern "C"
