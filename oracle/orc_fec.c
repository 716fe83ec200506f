/*
 * oracle/orc_fec.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 * CPU restatement of the reference's turbo FEC chain: code-block segmentation, QPP interleaver,
 * CRC, turbo encoder, turbo rate (de)matching and the three int16 max-log-MAP decoder numerics
 * (generic wrapping / 8-window ">>1" / 16-window) plus their iteration schedule.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ segmentation */

int orc_cb_index(uint32_t K)
{ /* cbsegm.c:115-126 */
  for (int j = 0; j < ORC_NOF_K; j++) {
    if (orc_qpp_table[j].K >= K) {
      return j;
    }
  }
  return -1;
}

int orc_cb_size(uint32_t idx) { return idx < ORC_NOF_K ? (int)orc_qpp_table[idx].K : -1; } /* cbsegm.c:133-139 */

int orc_cbsegm(orc_cbsegm_t* s, uint32_t tbs)
{ /* cbsegm.c:53-107 (36.212 5.1.2) */
  memset(s, 0, sizeof(*s));
  if (tbs == 0) {
    return 0;
  }
  uint32_t B = tbs + 24, Bp;
  s->tbs = tbs;
  if (B <= ORC_MAX_K) {
    s->C = 1;
    Bp   = B;
  } else {
    s->C = (uint32_t)ceilf((float)B / (ORC_MAX_K - 24));
    Bp   = B + 24 * s->C;
  }
  int idx1 = orc_cb_index((Bp - 1) / s->C + 1);
  if (idx1 < 0) {
    return -1;
  }
  s->K1     = (uint32_t)orc_cb_size((uint32_t)idx1);
  s->K1_idx = (uint32_t)idx1;
  if (s->C == 1) {
    s->C1 = 1;
  } else {
    s->K2     = (uint32_t)orc_cb_size((uint32_t)(idx1 > 0 ? idx1 - 1 : idx1));
    s->K2_idx = (uint32_t)idx1 - 1;
    s->C2     = (s->C * s->K1 - Bp) / (s->K1 - s->K2);
    s->C1     = s->C - s->C2;
  }
  s->F = s->C1 * s->K1 + s->C2 * s->K2 - Bp;
  return 0;
}

/* ------------------------------------------------------------------ QPP interleaver */

static inline uint32_t win_of_nat(uint32_t n, uint32_t K, uint32_t W) { return (n % (K / W)) * W + n / (K / W); }
static inline uint32_t nat_of_win(uint32_t x, uint32_t K, uint32_t W) { return (x % W) * (K / W) + x / W; }

int orc_qpp(uint32_t K, uint32_t W, uint16_t* fwd, uint16_t* rev)
{ /* tc_interl_lte.c:75-114: pi(i) = (f1 i + f2 i^2) mod K; W>1 re-indexes both tables into the
     window-interleaved domain x = k*W + w  <->  natural n = w*(K/W) + k */
  int idx = orc_cb_index(K);
  if (idx < 0 || orc_qpp_table[idx].K != K) {
    return -1;
  }
  uint64_t  f1 = orc_qpp_table[idx].f1, f2 = orc_qpp_table[idx].f2;
  uint16_t* f = malloc(K * sizeof(uint16_t));
  uint16_t* r = malloc(K * sizeof(uint16_t));
  for (uint64_t i = 0; i < K; i++) {
    uint64_t j = (f1 * i + f2 * i * i) % K;
    f[i]       = (uint16_t)j;
    r[j]       = (uint16_t)i;
  }
  if (W <= 1) {
    memcpy(fwd, f, K * sizeof(uint16_t));
    memcpy(rev, r, K * sizeof(uint16_t));
  } else {
    for (uint32_t i = 0; i < K; i++) {
      fwd[i] = (uint16_t)win_of_nat(f[nat_of_win(i, K, W)], K, W);
      rev[i] = (uint16_t)win_of_nat(r[nat_of_win(i, K, W)], K, W);
    }
  }
  free(f);
  free(r);
  return 0;
}

/* ------------------------------------------------------------------ CRC */

uint32_t orc_crc_bytes(uint32_t poly, int order, const uint8_t* data, int nbits)
{ /* crc.c:33-47,141-152: MSB-first, init 0, no reflection, no final xor; nbits multiple of 8 */
  uint64_t crc = 0, high = 1ull << (order - 1), mask = (1ull << order) - 1;
  for (int i = 0; i < nbits / 8; i++) {
    crc ^= ((uint64_t)data[i]) << (order - 8);
    for (int j = 0; j < 8; j++) {
      uint64_t bit = crc & high;
      crc <<= 1;
      if (bit) {
        crc ^= poly;
      }
    }
    crc &= mask;
  }
  return (uint32_t)crc;
}

/* ------------------------------------------------------------------ turbo encoder */

typedef struct { uint8_t r0, r1, r2; } rsc_t;
static inline uint8_t rsc_step(rsc_t* s, uint8_t bit)
{ /* turbocoder.c:120-125: feedback 1+D^2+D^3, feed-forward 1+D+D^3 */
  uint8_t in  = bit ^ (s->r2 ^ s->r1);
  uint8_t out = s->r2 ^ (s->r0 ^ in);
  s->r2       = s->r1;
  s->r1       = s->r0;
  s->r0       = in;
  return out;
}
static inline void rsc_tail(rsc_t* s, uint8_t* x, uint8_t* z)
{ /* turbocoder.c:151-165 */
  uint8_t bit = s->r2 ^ s->r1;
  *x          = bit;
  *z          = rsc_step(s, bit);
}

int orc_tcod_encode_bits(const uint8_t* in, uint8_t* out, uint32_t K)
{ /* turbocoder.c:76-186 (NULL-bit handling omitted: filler bits unsupported upstream, sch.c:194) */
  int idx = orc_cb_index(K);
  if (idx < 0 || orc_qpp_table[idx].K != K) {
    return -1;
  }
  uint16_t* f = malloc(K * 2);
  uint16_t* r = malloc(K * 2);
  orc_qpp(K, 1, f, r);
  rsc_t a = {0, 0, 0}, b = {0, 0, 0};
  for (uint32_t i = 0; i < K; i++) {
    out[3 * i]     = in[i];
    out[3 * i + 1] = rsc_step(&a, in[i]);
    out[3 * i + 2] = rsc_step(&b, in[f[i]]);
  }
  uint8_t* t = &out[3 * K];
  for (int j = 0; j < 3; j++) {
    rsc_tail(&a, &t[2 * j], &t[2 * j + 1]);
  }
  for (int j = 0; j < 3; j++) {
    rsc_tail(&b, &t[6 + 2 * j], &t[6 + 2 * j + 1]);
  }
  free(f);
  free(r);
  return 0;
}

static inline uint8_t getbit(const uint8_t* p, uint32_t i) { return (p[i >> 3] >> (7 - (i & 7))) & 1; }
static inline void    putbit(uint8_t* p, uint32_t i, uint8_t b)
{
  if (b) {
    p[i >> 3] |= (uint8_t)(0x80 >> (i & 7));
  } else {
    p[i >> 3] &= (uint8_t)~(0x80 >> (i & 7));
  }
}

int orc_tcod_encode_bytes(uint8_t* sys, uint8_t* par, uint32_t K)
{ /* turbocoder.c:189-371 output format: sys[K/8] = tail nibble of stream 0 in the high 4 bits;
     par = [p0: K bits | 4 tail bits of stream 1 | p1: K bits | 4 tail bits of stream 2], MSB first.
     CRC fusion (turbocoder.c:205-283) is done by the caller in this restatement (orc_dlsch_encode). */
  uint8_t* bits = calloc(K ? K : 1, 1);
  uint8_t* enc  = malloc(3 * K + 12);
  for (uint32_t i = 0; i < K; i++) {
    bits[i] = getbit(sys, i);
  }
  if (orc_tcod_encode_bits(bits, enc, K)) {
    free(bits);
    free(enc);
    return -1;
  }
  memset(par, 0, (2 * K + 8) / 8 + 1);
  sys[K / 8] = 0;
  for (uint32_t i = 0; i < K; i++) {
    putbit(par, i, enc[3 * i + 1]);
    putbit(par, K + 4 + i, enc[3 * i + 2]);
  }
  /* tail: 3 streams x 4 positions, stream j position i = tail[3i+j] (turbocoder.c:354-366) */
  for (uint32_t i = 0; i < 4; i++) {
    putbit(sys, K + i, enc[3 * K + 3 * i + 0]);
    putbit(par, K + i, enc[3 * K + 3 * i + 1]);
    putbit(par, 2 * K + 4 + i, enc[3 * K + 3 * i + 2]);
  }
  free(bits);
  free(enc);
  return (int)(3 * K + 12);
}

/* ------------------------------------------------------------------ rate matching (36.212 5.1.4.1) */

static const uint8_t RM_P[32] = {0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30,
                                 1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31};

/* circular buffer w of size 3*Kp: w2d[k] = index into d (3*i+s) or -1 for <NULL> */
static void rm_build_w(uint32_t K, int32_t* w2d, uint32_t* R_out, uint32_t* Kp_out)
{
  uint32_t D = K + 4, R = (D - 1) / 32 + 1, Kp = R * 32, ND = Kp - D;
  for (uint32_t j = 0; j < 32; j++) {
    for (uint32_t i = 0; i < R; i++) {
      int32_t y = (int32_t)(i * 32 + RM_P[j]) - (int32_t)ND; /* position in the stream, <0 dummy */
      uint32_t k = j * R + i;
      w2d[k]          = y >= 0 ? 3 * y + 0 : -1;
      w2d[Kp + 2 * k] = y >= 0 ? 3 * y + 1 : -1;
      int32_t y2 = (int32_t)((RM_P[k / R] + 32 * (k % R) + 1) % Kp) - (int32_t)ND;
      w2d[Kp + 2 * k + 1] = y2 >= 0 ? 3 * y2 + 2 : -1;
    }
  }
  *R_out  = R;
  *Kp_out = Kp;
}

int orc_rm_rx_table(uint32_t K, uint32_t rv, uint32_t W, uint16_t* table)
{ /* rm_turbo.c:160-233 (receive table) + :236-260 (SB re-layout when W = 8/16/32) */
  int32_t* w2d = malloc(3 * (K + 36) * sizeof(int32_t));
  uint32_t R, Kp;
  rm_build_w(K, w2d, &R, &Kp);
  uint32_t Ncb = 3 * Kp;
  uint32_t k0  = R * (2 * (uint32_t)ceilf((float)Ncb / (float)(8 * R)) * rv + 2);
  uint32_t n = 0, j = 0, out_len = 3 * K + 12;
  while (n < out_len) {
    int32_t d = w2d[(k0 + j) % Ncb];
    if (d >= 0) {
      uint32_t idx = (uint32_t)d;
      if (W > 1) {
        if (idx < 3 * K) {
          idx = (idx % 3) * (K + 32) + win_of_nat(idx / 3, K, W);
        } else {
          idx = (idx - 3 * K) + 3 * (K + 32);
        }
      }
      table[n++] = (uint16_t)idx;
    }
    j++;
  }
  free(w2d);
  return 0;
}

int orc_rm_turbo_rx(const int16_t* e, int16_t* w, uint32_t n_e, uint32_t K, uint32_t rv, uint32_t W)
{ /* rm_turbo.c:374-420: output[deinter[i % out_len]] += input[i], wrapping int16 */
  uint32_t  out_len = 3 * K + 12;
  uint16_t* t       = malloc(out_len * 2);
  orc_rm_rx_table(K, rv, W, t);
  for (uint32_t i = 0; i < n_e; i++) {
    w[t[i % out_len]] = (int16_t)(w[t[i % out_len]] + e[i]);
  }
  free(t);
  return 0;
}

int orc_rm_turbo_rx_8bit(const int8_t* e, int8_t* w, uint32_t n_e, uint32_t K, uint32_t rv, uint32_t W)
{ /* rm_turbo.c:422-465 scalar semantics */
  uint32_t  out_len = 3 * K + 12;
  uint16_t* t       = malloc(out_len * 2);
  orc_rm_rx_table(K, rv, W, t);
  for (uint32_t i = 0; i < n_e; i++) {
    w[t[i % out_len]] = (int8_t)(w[t[i % out_len]] + e[i]);
  }
  free(t);
  return 0;
}

int orc_rm_turbo_tx_bits(const uint8_t* d, uint8_t* e, uint32_t n_e, uint32_t K, uint32_t rv)
{ /* rm_turbo.c:328-372: sub-block interleave + circular-buffer bit selection */
  uint32_t  out_len = 3 * K + 12;
  uint16_t* t       = malloc(out_len * 2);
  orc_rm_rx_table(K, rv, 0, t);
  for (uint32_t i = 0; i < n_e; i++) {
    e[i] = d[t[i % out_len]];
  }
  free(t);
  return 0;
}

/* ------------------------------------------------------------------ turbo decoder */

#define TD_INF 10000

static inline int16_t sat16(int v) { return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }
static inline int16_t sadd(int16_t a, int16_t b) { return sat16((int)a + (int)b); }
static inline int16_t ssub(int16_t a, int16_t b) { return sat16((int)a - (int)b); }
static inline int16_t wadd(int16_t a, int16_t b) { return (int16_t)((int)a + (int)b); }
static inline int16_t smax(int16_t a, int16_t b) { return a > b ? a : b; }

uint32_t orc_tdec_autoimp_subblocks(uint32_t K)
{ /* turbodecoder.c:394-406 with LV_HAVE_AVX2 */
  if (!(K % 16) && K > 800) {
    return 16;
  } else if (!(K % 8) && K > 400) {
    return 8;
  }
  return 0;
}

uint32_t orc_tdec_autoimp_subblocks_8bit(uint32_t K)
{ /* turbodecoder.c:421-436 */
  if (!(K % 32) && K > 2048) {
    return 32;
  } else if (!(K % 16) && K > 800) {
    return 16;
  } else if (!(K % 8) && K > 400) {
    return 8;
  }
  return 0;
}

/* --- generic scalar decoder, wrapping int16 (turbodecoder_gen.c:54-233) */
static void gen_dec(const int16_t* input, const int16_t* app, const int16_t* parity, int16_t* output, uint32_t K, int16_t* beta)
{
  int16_t  m_b[8], nw[8], old[8];
  uint32_t end = K + 3;
  beta[8 * end] = 0;
  for (int i = 1; i < 8; i++) {
    beta[8 * end + i] = -TD_INF;
  }
  for (int i = 0; i < 8; i++) {
    old[i] = beta[8 * end + i];
  }
  for (int k = (int)end - 1; k >= 0; k--) {
    int16_t x = input[k];
    if (app && (uint32_t)k < K) {
      x = wadd(x, app[k]);
    }
    int16_t y = parity[k], xy = wadd(x, y);
    m_b[0] = wadd(old[4], xy); m_b[1] = old[4];           m_b[2] = wadd(old[5], y);  m_b[3] = wadd(old[5], x);
    m_b[4] = wadd(old[6], x);  m_b[5] = wadd(old[6], y);  m_b[6] = old[7];           m_b[7] = wadd(old[7], xy);
    nw[0] = old[0];            nw[1] = wadd(old[0], xy);  nw[2] = wadd(old[1], x);   nw[3] = wadd(old[1], y);
    nw[4] = wadd(old[2], y);   nw[5] = wadd(old[2], x);   nw[6] = wadd(old[3], xy);  nw[7] = old[3];
    for (int i = 0; i < 8; i++) {
      old[i]          = smax(m_b[i], nw[i]);
      beta[8 * k + i] = old[i];
    }
    if ((k % 4) == 0 && (uint32_t)k < K) {
      for (int i = 1; i < 8; i++) {
        old[i] = (int16_t)(old[i] - old[0]);
      }
      old[0] = 0;
    }
  }
  old[0] = 0;
  for (int i = 1; i < 8; i++) {
    old[i] = -TD_INF;
  }
  for (uint32_t k = 1; k < K + 1; k++) {
    int16_t x = input[k - 1];
    if (app) {
      x = wadd(x, app[k - 1]);
    }
    int16_t y = parity[k - 1], xy = wadd(x, y);
    m_b[0] = old[0];           m_b[1] = wadd(old[3], y);  m_b[2] = wadd(old[4], y);  m_b[3] = old[7];
    m_b[4] = old[1];           m_b[5] = wadd(old[2], y);  m_b[6] = wadd(old[5], y);  m_b[7] = old[6];
    nw[0] = wadd(old[1], xy);  nw[1] = wadd(old[2], x);   nw[2] = wadd(old[5], x);   nw[3] = wadd(old[6], xy);
    nw[4] = wadd(old[0], xy);  nw[5] = wadd(old[3], x);   nw[6] = wadd(old[4], x);   nw[7] = wadd(old[7], xy);
    int16_t m0 = wadd(m_b[0], beta[8 * k]), m1 = wadd(nw[0], beta[8 * k]);
    for (int i = 1; i < 8; i++) {
      m0 = smax(m0, wadd(m_b[i], beta[8 * k + i]));
      m1 = smax(m1, wadd(nw[i], beta[8 * k + i]));
    }
    for (int i = 0; i < 8; i++) {
      old[i] = smax(m_b[i], nw[i]);
    }
    if ((k % 4) == 0) {
      for (int i = 1; i < 8; i++) {
        old[i] = (int16_t)(old[i] - old[0]);
      }
      old[0] = 0;
    }
    output[k - 1] = (int16_t)(m1 - m0);
  }
}

/* --- windowed decoders, saturating int16 (turbodecoder_win.h:332-715; W=8: sse16, W=16: avx16) */
#define WIN_OVERLAP 40
#define WMAX 32

static inline void win_normalize(uint32_t k, int16_t old[8][WMAX], uint32_t W)
{ /* turbodecoder_win.h:332-349, normalize_period 2, subtract state 0 */
  if ((k % 2) == 0 && k != 0) {
    for (uint32_t w = 0; w < W; w++) {
      for (int i = 1; i < 8; i++) {
        old[i][w] = ssub(old[i][w], old[0][w]);
      }
      old[0][w] = 0;
    }
  }
}

static void win_beta_tail(const int16_t* input, const int16_t* parity, uint32_t K, int16_t old[8])
{ /* turbodecoder_win.h:351-395: scalar tail trellis; sadd() is a plain (wrapping) add for int16 */
  int16_t m_b[8], nw[8];
  old[0] = 0;
  for (int i = 1; i < 8; i++) {
    old[i] = -TD_INF;
  }
  for (int k = (int)K + 2; k >= (int)K; k--) {
    int16_t x = input[k], y = parity[k], xy = wadd(x, y);
    m_b[0] = wadd(old[4], xy); m_b[1] = old[4];           m_b[2] = wadd(old[5], y);  m_b[3] = wadd(old[5], x);
    m_b[4] = wadd(old[6], x);  m_b[5] = wadd(old[6], y);  m_b[6] = old[7];           m_b[7] = wadd(old[7], xy);
    nw[0] = old[0];            nw[1] = wadd(old[0], xy);  nw[2] = wadd(old[1], x);   nw[3] = wadd(old[1], y);
    nw[4] = wadd(old[2], y);   nw[5] = wadd(old[2], x);   nw[6] = wadd(old[3], xy);  nw[7] = old[3];
    for (int i = 0; i < 8; i++) {
      old[i] = smax(m_b[i], nw[i]);
    }
  }
}

static void win_dec(const int16_t* input, const int16_t* app, const int16_t* parity, int16_t* output, uint32_t K, uint32_t W,
                    int out_shift, int16_t* beta /* 8*(K/W+1)*W */)
{
  uint32_t L = K / W;
  int16_t  old[8][WMAX], m_b[8], nw[8];

  /* ---- beta (turbodecoder_win.h:398-526) */
  for (int pass = 0; pass < 2; pass++) {
    uint32_t len = pass == 0 ? WIN_OVERLAP : L;
    if (pass == 0) {
      for (int i = 0; i < 8; i++) {
        for (uint32_t w = 0; w < W; w++) {
          old[i][w] = -TD_INF;
        }
      }
    } else {
      int16_t tail[8];
      win_beta_tail(input, parity, K, tail);
      for (int i = 0; i < 8; i++) {
        for (uint32_t w = 0; w + 1 < W; w++) {
          old[i][w] = old[i][w + 1]; /* move_right: window w starts from the warm-up of window w+1 */
        }
        old[i][W - 1] = tail[i];
      }
      for (int i = 0; i < 8; i++) {
        for (uint32_t w = 0; w < W; w++) {
          beta[(8 * L + i) * W + w] = old[i][w];
        }
      }
    }
    for (int k = (int)len - 1; k >= 0; k--) {
      for (uint32_t w = 0; w < W; w++) {
        int16_t x = input[k * W + w], y = parity[k * W + w];
        if (app) {
          x = sadd(app[k * W + w], x);
        }
        int16_t xy = sadd(x, y);
        int16_t o[8];
        for (int i = 0; i < 8; i++) {
          o[i] = old[i][w];
        }
        m_b[0] = sadd(o[4], xy); m_b[1] = o[4];          m_b[2] = sadd(o[5], y);  m_b[3] = sadd(o[5], x);
        m_b[4] = sadd(o[6], x);  m_b[5] = sadd(o[6], y); m_b[6] = o[7];           m_b[7] = sadd(o[7], xy);
        nw[0] = o[0];            nw[1] = sadd(o[0], xy); nw[2] = sadd(o[1], x);   nw[3] = sadd(o[1], y);
        nw[4] = sadd(o[2], y);   nw[5] = sadd(o[2], x);  nw[6] = sadd(o[3], xy);  nw[7] = o[3];
        for (int i = 0; i < 8; i++) {
          old[i][w] = smax(m_b[i], nw[i]);
          if (pass == 1) {
            beta[(8 * k + i) * W + w] = old[i][w];
          }
        }
      }
      win_normalize((uint32_t)k, old, W);
    }
  }

  /* ---- alpha + output (turbodecoder_win.h:529-679) */
  for (int pass = 0; pass < 2; pass++) {
    uint32_t len = pass == 0 ? WIN_OVERLAP : L;
    if (pass == 0) {
      for (int i = 0; i < 8; i++) {
        for (uint32_t w = 0; w < W; w++) {
          old[i][w] = -TD_INF;
        }
      }
    } else {
      for (int i = 0; i < 8; i++) {
        for (uint32_t w = W - 1; w > 0; w--) {
          old[i][w] = old[i][w - 1]; /* move_left */
        }
        old[i][0] = i == 0 ? 0 : -TD_INF;
      }
    }
    uint32_t base = L - len;
    for (uint32_t k = 0; k < len; k++) {
      for (uint32_t w = 0; w < W; w++) {
        uint32_t p = (base + k) * W + w;
        int16_t  x = input[p], y = parity[p];
        if (app) {
          x = sadd(app[p], x);
        }
        int16_t xy = sadd(x, y);
        int16_t o[8];
        for (int i = 0; i < 8; i++) {
          o[i] = old[i][w];
        }
        m_b[0] = o[0];           m_b[1] = sadd(o[3], y); m_b[2] = sadd(o[4], y);  m_b[3] = o[7];
        m_b[4] = o[1];           m_b[5] = sadd(o[2], y); m_b[6] = sadd(o[5], y);  m_b[7] = o[6];
        nw[0] = sadd(o[1], xy);  nw[1] = sadd(o[2], x);  nw[2] = sadd(o[5], x);   nw[3] = sadd(o[6], xy);
        nw[4] = sadd(o[0], xy);  nw[5] = sadd(o[3], x);  nw[6] = sadd(o[4], x);   nw[7] = sadd(o[7], xy);
        if (pass == 1) {
          const int16_t* b  = &beta[(8 * (k + 1)) * W + w];
          int16_t        m0 = sadd(b[0], m_b[0]), m1 = sadd(b[0], nw[0]);
          for (int i = 1; i < 8; i++) {
            m0 = smax(m0, sadd(b[i * W], m_b[i]));
            m1 = smax(m1, sadd(b[i * W], nw[i]));
          }
          int16_t out = ssub(m1, m0);
          if (out_shift) {
            out = (int16_t)(out >> out_shift); /* arithmetic shift, _mm_srai_epi16 */
          }
          output[k * W + w] = out;
        }
        for (int i = 0; i < 8; i++) {
          old[i][w] = smax(m_b[i], nw[i]);
        }
      }
      win_normalize(k, old, W);
    }
  }
}

static void siso(const int16_t* in, const int16_t* app, const int16_t* par, int16_t* out, uint32_t K, uint32_t W, int16_t* beta)
{
  if (W == 0) {
    gen_dec(in, app, par, out, K, beta);
  } else {
    win_dec(in, app, par, out, K, W, W == 8 ? 1 : 0, beta); /* divide_output only for sse16 (win.h:56 vs 61-90) */
  }
}

static void decide(const int16_t* llr, uint8_t* out, uint32_t K, uint32_t W)
{ /* turbodecoder_gen.c:255-273 / turbodecoder_win.h:771-838: bit = llr > 0, MSB first, natural order */
  memset(out, 0, K / 8);
  for (uint32_t j = 0; j < K; j++) {
    int16_t v = W ? llr[win_of_nat(j, K, W)] : llr[j];
    if (v > 0) {
      out[j >> 3] |= (uint8_t)(0x80 >> (j & 7));
    }
  }
}

int orc_tdec_run_w(const int16_t* input, bool in_is_sb, uint32_t K, uint32_t W, uint32_t nof_iter, uint8_t* out, uint8_t* hard_per_iter)
{ /* schedule: turbodecoder_iter.h:71-139; front-end turbodecoder.c:383-390,497-562 */
  int idx = orc_cb_index(K);
  if (idx < 0 || orc_qpp_table[idx].K != K || (W && K % W) || (in_is_sb && !W)) {
    return -1;
  }
  uint32_t  len = K + 16;
  int16_t  *syst = calloc(len, 2), *par0 = calloc(len, 2), *par1 = calloc(len, 2);
  int16_t  *app1 = calloc(len, 2), *app2 = calloc(len, 2), *ext1 = calloc(len, 2), *ext2 = calloc(len, 2);
  int16_t*  beta = calloc(8 * (size_t)(K + 16) * (W ? 1 : 1) + 8 * WMAX, 2);
  uint16_t *inter = malloc(K * 2), *deinter = malloc(K * 2);
  orc_qpp(K, W ? W : 1, inter, deinter);

  /* input extraction (turbodecoder_gen.c:235-253, turbodecoder_win.h:727-769, turbodecoder_iter.h:58-68,84-91) */
  if (in_is_sb) {
    memcpy(syst, input, K * 2);
    memcpy(par0, input + (K + 32), K * 2);
    memcpy(par1, input + 2 * (K + 32), K * 2);
    for (uint32_t j = 0; j < 3; j++) {
      syst[K + j] = input[3 * (K + 32) + 2 * j];
      par0[K + j] = input[3 * (K + 32) + 2 * j + 1];
      app2[K + j] = input[3 * (K + 32) + 6 + 2 * j];
      par1[K + j] = input[3 * (K + 32) + 6 + 2 * j + 1];
    }
  } else {
    for (uint32_t n = 0; n < K; n++) {
      uint32_t x = W ? win_of_nat(n, K, W) : n;
      syst[x]    = input[3 * n];
      par0[x]    = input[3 * n + 1];
      par1[x]    = input[3 * n + 2];
    }
    for (uint32_t j = 0; j < 3; j++) {
      syst[K + j] = input[3 * K + 2 * j];
      par0[K + j] = input[3 * K + 2 * j + 1];
      app2[K + j] = input[3 * K + 6 + 2 * j];
      par1[K + j] = input[3 * K + 6 + 2 * j + 1];
    }
  }

  for (uint32_t n_iter = 0; n_iter < nof_iter; n_iter++) {
    if ((n_iter % 2) == 0) {
      if (n_iter) {
        for (uint32_t i = 0; i < K; i++) {
          app1[i] = (int16_t)(app1[i] - ext1[i]); /* srslte_vec_sub_sss: wrapping */
        }
      }
      siso(syst, n_iter ? app1 : NULL, par0, ext1, K, W, beta);
    } else {
      if (n_iter > 1) {
        for (uint32_t i = 0; i < K; i++) {
          ext1[i] = (int16_t)(ext1[i] - app1[i]);
        }
      }
      for (uint32_t i = 0; i < K; i++) {
        app2[deinter[i]] = ext1[i]; /* srslte_vec_lut_sss */
      }
      siso(app2, NULL, par1, ext2, K, W, beta);
      for (uint32_t i = 0; i < K; i++) {
        app1[inter[i]] = ext2[i];
      }
    }
    /* hard decision: n_iter (after increment) even -> app1, odd -> ext1 */
    const int16_t* src = ((n_iter + 1) % 2) == 0 ? app1 : ext1;
    if (hard_per_iter) {
      decide(src, &hard_per_iter[(size_t)n_iter * (K / 8)], K, W);
    }
    if (n_iter + 1 == nof_iter && out) {
      decide(src, out, K, W);
    }
  }
  free(syst); free(par0); free(par1); free(app1); free(app2); free(ext1); free(ext2); free(beta); free(inter); free(deinter);
  return 0;
}

int orc_tdec_run(const int16_t* input, bool in_is_sb, uint32_t K, uint32_t nof_iter, uint8_t* out, uint8_t* hard_per_iter)
{
  return orc_tdec_run_w(input, in_is_sb, K, orc_tdec_autoimp_subblocks(K), nof_iter, out, hard_per_iter);
}

/* --- 8-bit windowed decoders (turbodecoder_win.h with WINIMP sse8: W = 16, avx8: W = 32): saturating int8, INF = 0,
       max-normalisation after every step but the one with counter 0, extrinsic output >> 1 */
static inline int8_t sat8(int v) { return (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v)); }
static inline int8_t sadd8(int8_t a, int8_t b) { return sat8((int)a + (int)b); }
static inline int8_t ssub8(int8_t a, int8_t b) { return sat8((int)a - (int)b); }
static inline int8_t smax8(int8_t a, int8_t b) { return a > b ? a : b; }
static inline int8_t tail_sadd8(int8_t x, int8_t y)
{ /* turbodecoder_win.h:322-330 with use_saturated_add: clamps at +127 only, the negative side wraps in the cast */
  int16_t z = (int16_t)(x + y);
  return z > 127 ? 127 : (int8_t)z;
}

static inline void win_normalize8(uint32_t k, int8_t old[8][WMAX], uint32_t W)
{ /* turbodecoder_win.h:332-349 with normalize_max, normalize_period 1 */
  if (k != 0) {
    for (uint32_t w = 0; w < W; w++) {
      int8_t m = smax8(old[0][w], old[1][w]);
      for (int i = 2; i < 8; i++) {
        m = smax8(m, old[i][w]);
      }
      for (int i = 0; i < 8; i++) {
        old[i][w] = ssub8(old[i][w], m);
      }
    }
  }
}

static void win_beta_tail8(const int8_t* input, const int8_t* parity, uint32_t K, int8_t old[8])
{ /* turbodecoder_win.h:351-395 with INF = 0 */
  int8_t m_b[8], nw[8];
  for (int i = 0; i < 8; i++) {
    old[i] = 0;
  }
  for (int k = (int)K + 2; k >= (int)K; k--) {
    int8_t x = input[k], y = parity[k], xy = tail_sadd8(x, y);
#define TA tail_sadd8
    m_b[0] = TA(old[4], xy); m_b[1] = old[4];         m_b[2] = TA(old[5], y);  m_b[3] = TA(old[5], x);
    m_b[4] = TA(old[6], x);  m_b[5] = TA(old[6], y);  m_b[6] = old[7];         m_b[7] = TA(old[7], xy);
    nw[0] = old[0];          nw[1] = TA(old[0], xy);  nw[2] = TA(old[1], x);   nw[3] = TA(old[1], y);
    nw[4] = TA(old[2], y);   nw[5] = TA(old[2], x);   nw[6] = TA(old[3], xy);  nw[7] = old[3];
#undef TA
    for (int i = 0; i < 8; i++) {
      old[i] = m_b[i] > nw[i] ? m_b[i] : nw[i];
    }
  }
}

static void win_dec8(const int8_t* input, const int8_t* app, const int8_t* parity, int8_t* output, uint32_t K, uint32_t W,
                     int8_t* beta /* 8*(K/W+1)*W */)
{
  uint32_t L = K / W;
  int8_t   old[8][WMAX], m_b[8], nw[8];

  /* ---- beta (turbodecoder_win.h:398-526) */
  for (int pass = 0; pass < 2; pass++) {
    uint32_t len = pass == 0 ? WIN_OVERLAP : L;
    if (pass == 0) {
      memset(old, 0, sizeof(old)); /* -INF = 0 */
    } else {
      int8_t tail[8];
      win_beta_tail8(input, parity, K, tail);
      for (int i = 0; i < 8; i++) {
        for (uint32_t w = 0; w + 1 < W; w++) {
          old[i][w] = old[i][w + 1]; /* move_right (+ the 128-bit lane fix-up of avx8, :417-447) */
        }
        old[i][W - 1] = tail[i];
      }
      for (int i = 0; i < 8; i++) {
        for (uint32_t w = 0; w < W; w++) {
          beta[(8 * L + i) * W + w] = old[i][w];
        }
      }
    }
    for (int k = (int)len - 1; k >= 0; k--) {
      for (uint32_t w = 0; w < W; w++) {
        int8_t x = input[k * W + w], y = parity[k * W + w];
        if (app) {
          x = sadd8(app[k * W + w], x);
        }
        int8_t xy = sadd8(x, y);
        int8_t o[8];
        for (int i = 0; i < 8; i++) {
          o[i] = old[i][w];
        }
        m_b[0] = sadd8(o[4], xy); m_b[1] = o[4];           m_b[2] = sadd8(o[5], y);  m_b[3] = sadd8(o[5], x);
        m_b[4] = sadd8(o[6], x);  m_b[5] = sadd8(o[6], y); m_b[6] = o[7];            m_b[7] = sadd8(o[7], xy);
        nw[0] = o[0];             nw[1] = sadd8(o[0], xy); nw[2] = sadd8(o[1], x);   nw[3] = sadd8(o[1], y);
        nw[4] = sadd8(o[2], y);   nw[5] = sadd8(o[2], x);  nw[6] = sadd8(o[3], xy);  nw[7] = o[3];
        for (int i = 0; i < 8; i++) {
          old[i][w] = smax8(m_b[i], nw[i]);
          if (pass == 1) {
            beta[(8 * k + i) * W + w] = old[i][w];
          }
        }
      }
      win_normalize8((uint32_t)k, old, W);
    }
  }

  /* ---- alpha + output (turbodecoder_win.h:529-679) */
  for (int pass = 0; pass < 2; pass++) {
    uint32_t len = pass == 0 ? WIN_OVERLAP : L;
    if (pass == 0) {
      memset(old, 0, sizeof(old));
    } else {
      for (int i = 0; i < 8; i++) {
        for (uint32_t w = W - 1; w > 0; w--) {
          old[i][w] = old[i][w - 1]; /* move_left */
        }
        old[i][0] = 0; /* known state 0 and -INF are both 0 */
      }
    }
    uint32_t base = L - len;
    for (uint32_t k = 0; k < len; k++) {
      for (uint32_t w = 0; w < W; w++) {
        uint32_t p = (base + k) * W + w;
        int8_t   x = input[p], y = parity[p];
        if (app) {
          x = sadd8(app[p], x);
        }
        int8_t xy = sadd8(x, y);
        int8_t o[8];
        for (int i = 0; i < 8; i++) {
          o[i] = old[i][w];
        }
        m_b[0] = o[0];            m_b[1] = sadd8(o[3], y); m_b[2] = sadd8(o[4], y);  m_b[3] = o[7];
        m_b[4] = o[1];            m_b[5] = sadd8(o[2], y); m_b[6] = sadd8(o[5], y);  m_b[7] = o[6];
        nw[0] = sadd8(o[1], xy);  nw[1] = sadd8(o[2], x);  nw[2] = sadd8(o[5], x);   nw[3] = sadd8(o[6], xy);
        nw[4] = sadd8(o[0], xy);  nw[5] = sadd8(o[3], x);  nw[6] = sadd8(o[4], x);   nw[7] = sadd8(o[7], xy);
        if (pass == 1) {
          const int8_t* b  = &beta[(8 * (k + 1)) * W + w];
          int8_t        m0 = sadd8(b[0], m_b[0]), m1 = sadd8(b[0], nw[0]);
          for (int i = 1; i < 8; i++) {
            m0 = smax8(m0, sadd8(b[i * W], m_b[i]));
            m1 = smax8(m1, sadd8(b[i * W], nw[i]));
          }
          output[k * W + w] = (int8_t)(ssub8(m1, m0) >> 1); /* divide_output, simd_rb_shift (:143-147,:657-659) */
        }
        for (int i = 0; i < 8; i++) {
          old[i][w] = smax8(m_b[i], nw[i]);
        }
      }
      win_normalize8(k, old, W);
    }
  }
}

static void decide8(const int8_t* llr, uint8_t* out, uint32_t K, uint32_t W)
{
  memset(out, 0, K / 8);
  for (uint32_t j = 0; j < K; j++) {
    if (llr[win_of_nat(j, K, W)] > 0) {
      out[j >> 3] |= (uint8_t)(0x80 >> (j & 7));
    }
  }
}

static void vec_sub8(int8_t* x, const int8_t* y, uint32_t len)
{ /* srslte_vec_sub_bbb_simd (vector_simd.c:158-185), aligned buffers, AVX2 build: saturating in the 32-wide body,
     wrapping in the scalar tail */
  uint32_t body = len / 32 * 32;
  for (uint32_t i = 0; i < body; i++) {
    x[i] = ssub8(x[i], y[i]);
  }
  for (uint32_t i = body; i < len; i++) {
    x[i] = (int8_t)(x[i] - y[i]);
  }
}

int orc_tdec_run_8bit(const int8_t* input, bool in_is_sb, uint32_t K, uint32_t nof_iter, uint8_t* out, uint8_t* hard_per_iter)
{ /* turbodecoder.c:421-487,:565-593: AUTO selection for 8-bit LLRs on an AVX2 host. K > 2048 (K%32==0): avx8, W = 32;
     800 < K (K%16==0): sse8, W = 16; otherwise the LLRs are widened and a 16-bit back-end runs (:465-469).
     Upstream widens only 3K+12 elements even when the buffer is in the 3(K+32)+12 "SB" layout (:466), leaving the end of
     parity 1 and the tail LLRs to stale memory for 400 < K <= 800; here the whole buffer is widened. */
  uint32_t W = orc_tdec_autoimp_subblocks_8bit(K);
  int      idx = orc_cb_index(K);
  if (idx < 0 || orc_qpp_table[idx].K != K || (in_is_sb && !W)) {
    return -1;
  }
  if (W < 16) {
    uint32_t n    = in_is_sb ? 3 * (K + 32) + 12 : 3 * K + 12;
    int16_t* conv = malloc(n * 2);
    for (uint32_t i = 0; i < n; i++) {
      conv[i] = input[i];
    }
    int r = orc_tdec_run_w(conv, in_is_sb, K, W, nof_iter, out, hard_per_iter);
    free(conv);
    return r;
  }
  uint32_t  len = K + 16;
  int8_t   *syst = calloc(len, 1), *par0 = calloc(len, 1), *par1 = calloc(len, 1);
  int8_t   *app1 = calloc(len, 1), *app2 = calloc(len, 1), *ext1 = calloc(len, 1), *ext2 = calloc(len, 1);
  int8_t*   beta = calloc(8 * (size_t)(K + 16) + 8 * WMAX, 1);
  uint16_t *inter = malloc(K * 2), *deinter = malloc(K * 2);
  orc_qpp(K, W, inter, deinter);
  uint32_t tb = in_is_sb ? 3 * (K + 32) : 3 * K;
  if (in_is_sb) {
    memcpy(syst, input, K);
    memcpy(par0, input + (K + 32), K);
    memcpy(par1, input + 2 * (K + 32), K);
  } else {
    for (uint32_t n = 0; n < K; n++) {
      uint32_t x = win_of_nat(n, K, W);
      syst[x]    = input[3 * n];
      par0[x]    = input[3 * n + 1];
      par1[x]    = input[3 * n + 2];
    }
  }
  for (uint32_t j = 0; j < 3; j++) {
    syst[K + j] = input[tb + 2 * j];
    par0[K + j] = input[tb + 2 * j + 1];
    app2[K + j] = input[tb + 6 + 2 * j];
    par1[K + j] = input[tb + 6 + 2 * j + 1];
  }
  for (uint32_t n_iter = 0; n_iter < nof_iter; n_iter++) {
    if ((n_iter % 2) == 0) {
      if (n_iter) {
        vec_sub8(app1, ext1, K);
      }
      win_dec8(syst, n_iter ? app1 : NULL, par0, ext1, K, W, beta);
    } else {
      if (n_iter > 1) {
        vec_sub8(ext1, app1, K);
      }
      for (uint32_t i = 0; i < K; i++) {
        app2[deinter[i]] = ext1[i]; /* srslte_vec_lut_bbb */
      }
      win_dec8(app2, NULL, par1, ext2, K, W, beta);
      for (uint32_t i = 0; i < K; i++) {
        app1[inter[i]] = ext2[i];
      }
    }
    const int8_t* src = ((n_iter + 1) % 2) == 0 ? app1 : ext1;
    if (hard_per_iter) {
      decide8(src, &hard_per_iter[(size_t)n_iter * (K / 8)], K, W);
    }
    if (n_iter + 1 == nof_iter && out) {
      decide8(src, out, K, W);
    }
  }
  free(syst); free(par0); free(par1); free(app1); free(app2); free(ext1); free(ext2); free(beta); free(inter); free(deinter);
  return 0;
}

/* ------------------------------------------------------------------ DL-SCH (sch.c) */

int orc_dlsch_encode(const orc_sch_cfg_t* cfg, const uint8_t* data, uint8_t* e_bits)
{ /* sch.c:183-289 with rv = cfg->rv; CRC fusion of turbocoder.c:205-283 made explicit */
  orc_cbsegm_t s;
  if (orc_cbsegm(&s, cfg->tbs) || s.F) {
    return -1;
  }
  uint32_t tb_bytes = cfg->tbs / 8;
  uint8_t* tb       = malloc(tb_bytes + 3);
  memcpy(tb, data, tb_bytes);
  uint32_t crc = orc_crc_bytes(ORC_CRC24A, 24, data, (int)cfg->tbs);
  tb[tb_bytes] = (uint8_t)(crc >> 16); tb[tb_bytes + 1] = (uint8_t)(crc >> 8); tb[tb_bytes + 2] = (uint8_t)crc;

  uint32_t Gp = cfg->nof_bits / cfg->Qm, gamma = Gp % s.C, rp = 0, wp = 0;
  uint8_t *cb = malloc(ORC_MAX_K / 8 + 4), *bits = malloc(ORC_MAX_K), *enc = malloc(3 * ORC_MAX_K + 12);
  for (uint32_t i = 0; i < s.C; i++) {
    uint32_t K    = i < s.C2 ? s.K2 : s.K1; /* encoder order: K- blocks first (sch.c:220-226) */
    uint32_t rlen = s.C > 1 ? K - 24 : K;
    uint32_t n_e  = (i <= s.C - gamma - 1) ? cfg->Qm * (Gp / s.C) : cfg->Qm * (uint32_t)ceilf((float)Gp / s.C);
    memcpy(cb, &tb[rp / 8], rlen / 8);
    if (s.C > 1) {
      uint32_t c = orc_crc_bytes(ORC_CRC24B, 24, cb, (int)rlen);
      cb[rlen / 8] = (uint8_t)(c >> 16); cb[rlen / 8 + 1] = (uint8_t)(c >> 8); cb[rlen / 8 + 2] = (uint8_t)c;
    }
    for (uint32_t b = 0; b < K; b++) {
      bits[b] = getbit(cb, b);
    }
    orc_tcod_encode_bits(bits, enc, K);
    orc_rm_turbo_tx_bits(enc, &e_bits[wp], n_e, K, cfg->rv);
    rp += rlen;
    wp += n_e;
  }
  free(tb); free(cb); free(bits); free(enc);
  return 0;
}

#define ORC_SB_STRIDE (3 * (ORC_MAX_K + 32) + 12) /* int16 per code block of a HARQ soft buffer (softbuffer.h:50 reserves 18600) */

static int dlsch_decode(const orc_sch_cfg_t* cfg, const void* e_any, bool llr8, uint8_t* data, uint32_t* cb_iters, uint8_t* cb_crc_ok,
                        int16_t* sb, uint8_t* sb_crc, uint8_t* sb_data, bool new_data)
{ /* sch.c:299-414 (decode_tb_cb) + :429-500 (decode_tb); llr8: the q->llr_is_8bit branches. sb == NULL: first transmission into a zeroed
     soft buffer. Otherwise sb / sb_crc / sb_data are the srslte_softbuffer_rx_t of this transport block (buffer_f, cb_crc, data;
     softbuffer.c:46-150) and new_data says whether the MAC reset it (srslte_softbuffer_rx_reset_tbs) before this call. */
  const int16_t* e = e_any;
  const int8_t*  e8 = e_any;
  orc_cbsegm_t s;
  if (orc_cbsegm(&s, cfg->tbs) || s.F) {
    return -2;
  }
  data[cfg->tbs / 8] = data[cfg->tbs / 8 + 1] = data[cfg->tbs / 8 + 2] = 0;
  int16_t* wtmp = malloc(ORC_SB_STRIDE * 2);
  uint8_t* hard = malloc(ORC_MAX_K / 8);
  bool     all_ok = true;
  if (sb && new_data) memset(sb_crc, 0, s.C);
  for (uint32_t cb = 0; cb < s.C; cb++) {
    uint32_t K    = cb < s.C1 ? s.K1 : s.K2; /* decoder order quirk: K+ blocks first (sch.c:320-321) */
    uint32_t rlen = s.C == 1 ? K : K - 24;
    if (sb && sb_crc[cb]) { /* "Do not process blocks with CRC Ok" (sch.c:317-318,:392-396) */
      memcpy(&data[cb * rlen / 8], &sb_data[cb * (ORC_MAX_K / 8)], rlen / 8);
      if (cb_iters) cb_iters[cb] = 0;
      if (cb_crc_ok) cb_crc_ok[cb] = 1;
      continue;
    }
    int16_t* w = sb ? sb + (size_t)cb * ORC_SB_STRIDE : wtmp;
    uint32_t Gp = cfg->nof_bits / cfg->Qm, gamma = Gp % s.C, n_e = cfg->Qm * (Gp / s.C), rp = cb * n_e, n_e2 = n_e;
    if (cb > s.C - gamma) { /* quirk: '>' where the encoder uses '>=' (sch.c:331-334 vs :232-236) */
      n_e2 = n_e + cfg->Qm;
      rp   = (s.C - gamma) * n_e + (cb - (s.C - gamma)) * n_e2;
    }
    uint32_t W = llr8 ? orc_tdec_autoimp_subblocks_8bit(K) : orc_tdec_autoimp_subblocks(K);
    if (!sb || new_data) memset(w, 0, (3 * (K + 32) + 12) * 2);
    bool     ok  = false;
    uint32_t noi = 0;
    /* hard decisions after each pass are independent of later passes, so run them one at a time */
    uint8_t* per = malloc((size_t)cfg->max_iter * (K / 8));
    if (llr8) {
      orc_rm_turbo_rx_8bit(&e8[rp], (int8_t*)w, n_e2, K, cfg->rv, W);
      orc_tdec_run_8bit((int8_t*)w, W != 0, K, cfg->max_iter, hard, per);
    } else {
      orc_rm_turbo_rx(&e[rp], w, n_e2, K, cfg->rv, W);
      orc_tdec_run_w(w, W != 0, K, W, cfg->max_iter, hard, per);
    }
    do {
      memcpy(&data[cb * rlen / 8], &per[(size_t)noi * (K / 8)], K / 8);
      noi++;
      uint32_t c = s.C > 1 ? orc_crc_bytes(ORC_CRC24B, 24, &data[cb * rlen / 8], (int)K)
                           : orc_crc_bytes(ORC_CRC24A, 24, &data[cb * rlen / 8], (int)(s.tbs + 24));
      ok = c == 0;
    } while (noi < cfg->max_iter && !ok);
    free(per);
    if (cb_iters) {
      cb_iters[cb] = noi;
    }
    if (cb_crc_ok) {
      cb_crc_ok[cb] = ok;
    }
    all_ok = all_ok && ok;
    if (sb) sb_crc[cb] = ok;
  }
  free(wtmp);
  free(hard);
  if (!all_ok) {
    if (sb) { /* save the blocks that passed for the next retransmission (sch.c:404-412) */
      for (uint32_t cb = 0; cb < s.C; cb++) {
        uint32_t K = cb < s.C1 ? s.K1 : s.K2, rlen = s.C == 1 ? K : K - 24;
        if (sb_crc[cb]) memcpy(&sb_data[cb * (ORC_MAX_K / 8)], &data[cb * rlen / 8], rlen / 8);
      }
    }
    return -1;
  }
  uint32_t par_rx = orc_crc_bytes(ORC_CRC24A, 24, data, (int)cfg->tbs);
  uint32_t par_tx = ((uint32_t)data[cfg->tbs / 8] << 16) | ((uint32_t)data[cfg->tbs / 8 + 1] << 8) | data[cfg->tbs / 8 + 2];
  return (par_rx == par_tx && par_rx) ? 0 : -1;
}

int orc_dlsch_decode(const orc_sch_cfg_t* cfg, const int16_t* e, uint8_t* data, uint32_t* cb_iters, uint8_t* cb_crc_ok)
{
  return dlsch_decode(cfg, e, false, data, cb_iters, cb_crc_ok, NULL, NULL, NULL, true);
}

int orc_dlsch_decode_8bit(const orc_sch_cfg_t* cfg, const int8_t* e, uint8_t* data, uint32_t* cb_iters, uint8_t* cb_crc_ok)
{
  return dlsch_decode(cfg, e, true, data, cb_iters, cb_crc_ok, NULL, NULL, NULL, true);
}

uint32_t orc_harq_softbuffer_stride(void) { return ORC_SB_STRIDE; }

int orc_dlsch_decode_harq(const orc_sch_cfg_t* cfg, const void* e, int llr8, int new_data, int16_t* softbuf, uint8_t* sb_cb_crc, uint8_t* sb_data,
                          uint8_t* data, uint32_t* cb_iters, uint8_t* cb_crc_ok)
{
  return dlsch_decode(cfg, e, llr8 != 0, data, cb_iters, cb_crc_ok, softbuf, sb_cb_crc, sb_data, new_data != 0);
}
