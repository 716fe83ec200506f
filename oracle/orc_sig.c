/*
 * oracle/orc_sig.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 * CPU restatement of: LTE numerology (phy_common.c), Gold sequence (sequence.c), CRS generation
 * and mapping (refsignal_dl.c), DFT (the transform FFTW computes for dft_fftw.c), OFDM
 * modulator/demodulator (ofdm.c, guru path) and SC-FDMA transform precoding (dft_precoding.c).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ numerology */

static bool use_standard_rates = false; /* phy_common.c:292-299: a process-wide switch, as upstream */
void orc_use_standard_symbol_size(bool enabled) { use_standard_rates = enabled; }

int orc_symbol_sz(int nof_prb)
{ /* phy_common.c:322-345: the default table, or the power-of-two family after srslte_use_standard_symbol_size(true) */
  if (nof_prb <= 0) return -1;
  if (use_standard_rates) return orc_symbol_sz_power2(nof_prb);
  if (nof_prb <= 6) return 128;
  if (nof_prb <= 15) return 256;
  if (nof_prb <= 25) return 384;
  if (nof_prb <= 50) return 768;
  if (nof_prb <= 75) return 1024;
  if (nof_prb <= 110) return 1536;
  return -1;
}

static int cp_len(int N, int c) { return (c * N + 2047) / 2048; } /* SRSLTE_CP_LEN: ceil(c*N/2048), phy_common.h:93-99 */
int orc_cp_len_norm(int sym_in_slot, int N) { return sym_in_slot == 0 ? cp_len(N, 160) : cp_len(N, 144); }
int orc_cp_len_ext(int N) { return cp_len(N, 512); }

/* ------------------------------------------------------------------ Gold sequence */

void orc_gold(uint32_t c_init, uint32_t len, uint8_t* c)
{ /* sequence.c:48-79 (36.211 7.2): Nc = 1600, x1(0)=1, x2 = c_init bits */
  const uint32_t Nc = 1600;
  uint8_t* x1 = calloc(Nc + len + 31, 1);
  uint8_t* x2 = calloc(Nc + len + 31, 1);
  for (int n = 0; n < 31; n++) {
    x2[n] = (c_init >> n) & 1;
  }
  x1[0] = 1;
  for (uint32_t n = 0; n < Nc + len; n++) {
    x1[n + 31] = (x1[n + 3] + x1[n]) & 1;
    x2[n + 31] = (x2[n + 3] + x2[n + 2] + x2[n + 1] + x2[n]) & 1;
  }
  for (uint32_t n = 0; n < len; n++) {
    c[n] = (x1[n + Nc] + x2[n + Nc]) & 1;
  }
  free(x1);
  free(x2);
}

/* ------------------------------------------------------------------ CRS */

static uint32_t crs_v(uint32_t port_id, uint32_t l)
{ /* refsignal_dl.c:134-168 */
  switch (port_id) {
    case 0: return (l % 2) ? 3 : 0;
    case 1: return (l % 2) ? 0 : 3;
    case 2: return l == 0 ? 0 : 3;
    default: return l == 0 ? 3 : 0;
  }
}
static uint32_t crs_nof_symbols(uint32_t port_id) { return port_id < 2 ? 4 : 2; } /* refsignal_dl.c:170-177 FDD */
uint32_t orc_crs_nsymbol(uint32_t l, bool cp_norm, uint32_t port_id)
{ /* refsignal_dl.c:234-249 */
  uint32_t nsymb = cp_norm ? 7 : 6;
  if (port_id < 2) {
    return (l % 2) ? (l / 2 + 1) * nsymb - 3 : (l / 2) * nsymb;
  }
  return 1 + l * nsymb;
}
uint32_t orc_crs_fidx(const orc_cell_t* cell, uint32_t l, uint32_t port_id) { return (crs_v(port_id, l) + (cell->id % 6)) % 6; }

int orc_crs_pilots(const orc_cell_t* cell, uint32_t sf_idx, uint32_t port_id, orc_cf_t* pilots)
{ /* refsignal_dl.c:66-116: c_init = 1024(7(ns+1)+l'+1)(2 N_id+1) + 2 N_id + N_cp; offset MAX_PRB - nof_prb */
  const uint32_t MAX_PRB = 110;
  uint32_t       p = port_id / 2, nsym_slot = crs_nof_symbols(2 * p) / 2, N_cp = cell->cp_norm ? 1 : 0;
  uint8_t*       c = malloc(4 * MAX_PRB);
  for (uint32_t s = 0; s < 2; s++) {
    uint32_t ns = 2 * sf_idx + s;
    for (uint32_t l = 0; l < nsym_slot; l++) {
      uint32_t lp     = orc_crs_nsymbol(l, cell->cp_norm, 2 * p);
      uint32_t c_init = 1024 * (7 * (ns + 1) + lp + 1) * (2 * cell->id + 1) + 2 * cell->id + N_cp;
      orc_gold(c_init, 4 * MAX_PRB, c);
      for (uint32_t i = 0; i < 2 * cell->nof_prb; i++) {
        uint32_t  mp  = i + MAX_PRB - cell->nof_prb;
        orc_cf_t* dst = &pilots[2 * cell->nof_prb * (s * nsym_slot + l) + i];
        dst->re       = (float)((1 - 2 * (float)c[2 * mp]) / sqrt(2));
        dst->im       = (float)((1 - 2 * (float)c[2 * mp + 1]) / sqrt(2));
      }
    }
  }
  free(c);
  return 0;
}

/* TDD frame structure (36.211 Tables 4.2-1 and 4.2-2 as phy_common.c:85-99 holds them) */
int orc_tdd_sf_type(const orc_cell_t* cell, uint32_t sf_idx)
{ /* srslte_sfidx_tdd_type, phy_common.c:101-108 */
  static const char* const cfgs[7] = {"DSUUUDSUUU", "DSUUDDSUUD", "DSUDDDSUDD", "DSUUUDDDDD", "DSUUDDDDDD", "DSUDDDDDDD", "DSUUUDSUUD"};
  if (!cell->frame_type || cell->tdd_sf_config > 6 || sf_idx > 9) return 0;
  const char c = cfgs[cell->tdd_sf_config][sf_idx];
  return c == 'D' ? 0 : (c == 'U' ? 1 : 2);
}
uint32_t orc_tdd_nof_dw(const orc_cell_t* cell)
{ /* srslte_sfidx_tdd_nof_dw, phy_common.c:128-135: the table's first column, whatever the cyclic prefix */
  static const uint32_t dw[10] = {3, 9, 10, 11, 12, 3, 9, 10, 11, 6};
  return cell->tdd_ss_config < 10 ? dw[cell->tdd_ss_config] : 0;
}
uint32_t orc_nof_symb_slot(const orc_cell_t* cell, uint32_t sf_idx, uint32_t slot)
{ /* srslte_ra_dl_compute_nof_re (ra_dl.c:446-460) with srslte_sfidx_tdd_nof_dw_slot (phy_common.c:110-126) */
  const uint32_t nsymb = cell->cp_norm ? 7 : 6;
  if (orc_tdd_sf_type(cell, sf_idx) != 2) return nsymb;
  const uint32_t n = orc_tdd_nof_dw(cell);
  if (n < nsymb) return slot == 1 ? 0 : n;
  return slot == 1 ? n - nsymb : nsymb;
}
uint32_t orc_crs_nof_symbols(const orc_cell_t* cell, uint32_t sf_idx, uint32_t port_id)
{ /* srslte_refsignal_cs_nof_symbols, refsignal_dl.c:162-225 */
  if (orc_tdd_sf_type(cell, sf_idx) == 0 || !cell->frame_type) return port_id < 2 ? 4 : 2;
  const uint32_t dw = orc_tdd_nof_dw(cell);
  const uint32_t t3 = cell->cp_norm ? 12 : 10, t2 = cell->cp_norm ? 9 : 8, t1 = cell->cp_norm ? 5 : 4;
  if (dw >= t3) return port_id < 2 ? 4 : 2;
  if (dw >= t2) return port_id < 2 ? 3 : 2;
  if (dw >= t1) return port_id < 2 ? 2 : 1;
  return 1;
}

int orc_crs_put_sf(const orc_cell_t* cell, uint32_t sf_idx, uint32_t port_id, orc_cf_t* grid)
{ /* refsignal_dl.c:253-272: the CRS symbols the subframe has (all of a port's outside TDD special subframes) */
  uint32_t  nsym   = orc_crs_nof_symbols(cell, sf_idx, port_id), nre = 12 * cell->nof_prb;
  orc_cf_t* pilots = malloc(sizeof(orc_cf_t) * crs_nof_symbols(port_id) * 2 * cell->nof_prb);
  orc_crs_pilots(cell, sf_idx, port_id, pilots);
  for (uint32_t l = 0; l < nsym; l++) {
    uint32_t sym = orc_crs_nsymbol(l, cell->cp_norm, port_id), fidx = orc_crs_fidx(cell, l, port_id);
    for (uint32_t i = 0; i < 2 * cell->nof_prb; i++) {
      grid[sym * nre + fidx + 6 * i] = pilots[2 * cell->nof_prb * l + i];
    }
  }
  free(pilots);
  return 0;
}

/* ------------------------------------------------------------------ DFT */

void orc_dft_exact(const orc_cf_t* in, orc_cf_t* out, int N, int forward)
{ /* the transform fftwf_execute computes (unnormalised, sign -1 forward / +1 backward),
     evaluated directly in double precision: the pin for every FFT in this repo */
  double  sgn = forward ? -1.0 : 1.0;
  double* cs  = malloc(sizeof(double) * 2 * N);
  for (int k = 0; k < N; k++) {
    cs[2 * k]     = cos(2.0 * M_PI * k / N);
    cs[2 * k + 1] = sgn * sin(2.0 * M_PI * k / N);
  }
  for (int k = 0; k < N; k++) {
    double re = 0, im = 0;
    for (int n = 0; n < N; n++) {
      int    t = (int)(((long long)k * n) % N);
      double c = cs[2 * t], s = cs[2 * t + 1];
      re += in[n].re * c - in[n].im * s;
      im += in[n].re * s + in[n].im * c;
    }
    out[k].re = (float)re;
    out[k].im = (float)im;
  }
  free(cs);
}

/* Stockham mixed-radix FFT (radices 4,2,3,5); float data, double-derived twiddles. CPU baseline FFT. */
typedef struct { int N, nf, radix[16]; orc_cf_t* tw; } fft_plan_t;
#define PLAN_CACHE 64
static fft_plan_t g_plans[PLAN_CACHE];
static int        g_nplans = 0;

static fft_plan_t* fft_get_plan(int N)
{
  for (int i = 0; i < g_nplans; i++) {
    if (g_plans[i].N == N) return &g_plans[i];
  }
  if (g_nplans == PLAN_CACHE) return NULL;
  fft_plan_t p;
  p.N = N;
  p.nf = 0;
  int n = N;
  while (n % 4 == 0) { p.radix[p.nf++] = 4; n /= 4; }
  while (n % 2 == 0) { p.radix[p.nf++] = 2; n /= 2; }
  while (n % 3 == 0) { p.radix[p.nf++] = 3; n /= 3; }
  while (n % 5 == 0) { p.radix[p.nf++] = 5; n /= 5; }
  if (n != 1) return NULL;
  p.tw = malloc(sizeof(orc_cf_t) * N);
  for (int k = 0; k < N; k++) {
    p.tw[k].re = (float)cos(2.0 * M_PI * k / N);
    p.tw[k].im = (float)-sin(2.0 * M_PI * k / N);
  }
  g_plans[g_nplans] = p;
  return &g_plans[g_nplans++];
}

static inline orc_cf_t cmul(orc_cf_t a, orc_cf_t b) { return (orc_cf_t){a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
static inline orc_cf_t cadd(orc_cf_t a, orc_cf_t b) { return (orc_cf_t){a.re + b.re, a.im + b.im}; }
static inline orc_cf_t csub(orc_cf_t a, orc_cf_t b) { return (orc_cf_t){a.re - b.re, a.im - b.im}; }
static inline orc_cf_t cmulj(orc_cf_t a, float s) { return (orc_cf_t){-s * a.im, s * a.re}; } /* a * (j*s) */

static void butterfly(orc_cf_t* v, int R, float sgn /* -1 fwd, +1 bwd */)
{
  if (R == 2) {
    orc_cf_t a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  } else if (R == 4) {
    orc_cf_t a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = cmulj(csub(v[1], v[3]), sgn);
    v[0] = cadd(a, c); v[1] = cadd(b, d); v[2] = csub(a, c); v[3] = csub(b, d);
  } else if (R == 3) {
    const float c = -0.5f, s = 0.86602540378443864676f;
    orc_cf_t t = cadd(v[1], v[2]), u = cmulj(csub(v[1], v[2]), sgn * s);
    orc_cf_t m = {v[0].re + c * t.re, v[0].im + c * t.im};
    v[0] = cadd(v[0], t); v[1] = cadd(m, u); v[2] = csub(m, u);
  } else { /* 5 */
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f, s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    orc_cf_t t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), d1 = csub(v[1], v[4]), d2 = csub(v[2], v[3]);
    orc_cf_t m1 = {v[0].re + c1 * t1.re + c2 * t2.re, v[0].im + c1 * t1.im + c2 * t2.im};
    orc_cf_t m2 = {v[0].re + c2 * t1.re + c1 * t2.re, v[0].im + c2 * t1.im + c1 * t2.im};
    orc_cf_t u1 = cmulj((orc_cf_t){s1 * d1.re + s2 * d2.re, s1 * d1.im + s2 * d2.im}, sgn);
    orc_cf_t u2 = cmulj((orc_cf_t){s2 * d1.re - s1 * d2.re, s2 * d1.im - s1 * d2.im}, sgn);
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = cadd(m1, u1); v[4] = csub(m1, u1); v[2] = cadd(m2, u2); v[3] = csub(m2, u2);
  }
}

int orc_fft(const orc_cf_t* in, orc_cf_t* out, int N, int forward)
{
  fft_plan_t* p = fft_get_plan(N);
  if (!p) return -1;
  orc_cf_t* a = malloc(sizeof(orc_cf_t) * N);
  orc_cf_t* b = malloc(sizeof(orc_cf_t) * N);
  memcpy(a, in, sizeof(orc_cf_t) * N);
  float sgn = forward ? -1.f : 1.f;
  int   Ns  = 1;
  for (int f = 0; f < p->nf; f++) {
    int R = p->radix[f], nb = N / R;
    for (int j = 0; j < nb; j++) {
      int      k = j % Ns;
      orc_cf_t v[5] = {{0, 0}};
      for (int r = 0; r < R; r++) {
        orc_cf_t w = p->tw[((long long)r * k * (N / (Ns * R))) % N];
        if (!forward) w.im = -w.im;
        v[r] = cmul(a[j + r * nb], w);
      }
      butterfly(v, R, sgn);
      int j0 = (j / Ns) * Ns * R + k;
      for (int r = 0; r < R; r++) {
        b[j0 + r * Ns] = v[r];
      }
    }
    Ns *= R;
    orc_cf_t* t = a; a = b; b = t;
  }
  memcpy(out, a, sizeof(orc_cf_t) * N);
  free(a);
  free(b);
  return 0;
}

static void dft_any(const orc_cf_t* in, orc_cf_t* out, int N, int forward, bool exact)
{
  if (exact || orc_fft(in, out, N, forward)) {
    orc_dft_exact(in, out, N, forward);
  }
}

void orc_dft_r2hc(const float* in, float* out, int N, int forward)
{ /* dft_fftw.c:209-232 plans FFTW_R2HC (forward) / FFTW_HC2R (backward); FFTW's half-complex layout is
     r0 r1 .. r[N/2] i[(N+1)/2-1] .. i1, the backward transform is unnormalised */
  orc_cf_t *a = calloc((size_t)N, sizeof(orc_cf_t)), *b = calloc((size_t)N, sizeof(orc_cf_t));
  if (forward) {
    for (int i = 0; i < N; i++) a[i].re = in[i];
    orc_dft_exact(a, b, N, 1);
    for (int k = 0; k <= N / 2; k++) out[k] = b[k].re;
    for (int k = 1; k < N - k; k++) out[N - k] = b[k].im;
  } else {
    a[0].re = in[0];
    for (int k = 1; k < N - k; k++) {
      a[k]     = (orc_cf_t){in[k], in[N - k]};
      a[N - k] = (orc_cf_t){in[k], -in[N - k]};
    }
    if (N % 2 == 0) a[N / 2].re = in[N / 2];
    orc_dft_exact(a, b, N, 0);
    for (int i = 0; i < N; i++) out[i] = b[i].re;
  }
  free(a);
  free(b);
}

/* ------------------------------------------------------------------ OFDM */

int orc_symbol_sz_power2(int nof_prb)
{ /* phy_common.c:304-320: the sizes srslte_symbol_sz returns after srslte_use_standard_symbol_size(true) */
  if (nof_prb <= 0) return -1;
  if (nof_prb <= 6) return 128;
  if (nof_prb <= 15) return 256;
  if (nof_prb <= 25) return 512;
  if (nof_prb <= 50) return 1024;
  if (nof_prb <= 75) return 1536;
  if (nof_prb <= 110) return 2048;
  return -1;
}

int orc_ofdm_init(orc_ofdm_t* q, int nof_prb, bool cp_norm) { return orc_ofdm_init_sz(q, nof_prb, orc_symbol_sz(nof_prb), cp_norm); }

int orc_ofdm_init_sz(orc_ofdm_t* q, int nof_prb, int symbol_sz, bool cp_norm)
{ /* ofdm.c:38-57: srslte_ofdm_init_ takes the symbol size from its caller (srslte_symbol_sz of either rate family, ofdm.c:235-273) */
  memset(q, 0, sizeof(*q));
  int N = symbol_sz;
  if (N < 0 || nof_prb <= 0 || 12 * nof_prb >= N) return -1;
  q->nof_prb = nof_prb; q->symbol_sz = N; q->nof_re = 12 * nof_prb; q->nof_symbols = cp_norm ? 7 : 6;
  q->sf_sz = 15 * N; q->slot_sz = 15 * N / 2; q->cp_norm = cp_norm;
  return 0;
}

static orc_cf_t shift_val(const orc_ofdm_t* q, int t, int cplen)
{ /* ofdm.c:360-378: cexpf(I*2*pi*(t-cplen)*freq_shift/N) */
  double ph = 2.0 * M_PI * ((float)t - (float)cplen) * q->freq_shift_f / q->symbol_sz;
  return (orc_cf_t){(float)cos(ph), (float)sin(ph)};
}

static void sym_layout(const orc_ofdm_t* q, int s, int* pos_out, int* cpl_out)
{ /* first sample and CP length of symbol s. Regular: ofdm.c:384-393. MBSFN subframe (rx ofdm.c:424-437, tx :558-574): slot 0
     = non_mbsfn_region normal-CP symbols, a guard of SRSLTE_NON_MBSFN_REGION_GUARD_LENGTH (phy_common.h:147), then
     extended-CP symbols; slot 1 is a plain extended-CP slot (ofdm.c:463-465,:588-590). */
  int N = q->symbol_sz, pos = 0, cpl = 0, reg = q->non_mbsfn_region;
  for (int i = 0; i <= s; i++) {
    int l = i % q->nof_symbols;
    pos += i ? cpl + N : 0;
    if (reg && i < q->nof_symbols) {
      if (i == reg) {
        pos += reg == 1 ? orc_cp_len_ext(N) - orc_cp_len_norm(0, N) : 2 * orc_cp_len_ext(N) - orc_cp_len_norm(0, N) - orc_cp_len_norm(1, N);
      }
      cpl = i >= reg ? orc_cp_len_ext(N) : orc_cp_len_norm(i, N);
    } else {
      cpl = q->cp_norm ? orc_cp_len_norm(l, N) : orc_cp_len_ext(N);
    }
  }
  *pos_out = pos;
  *cpl_out = cpl;
}

void orc_ofdm_rx_sf(const orc_ofdm_t* q, const orc_cf_t* in_time, orc_cf_t* out_grid)
{ /* ofdm.c:398-422 (rx_slot, guru path) + :453-457 (time-domain shift); dc = !freq_shift (ofdm.c:374) */
  int       N = q->symbol_sz, nre = q->nof_re, dc = q->freq_shift ? 0 : 1;
  float     norm = 1.0f / sqrtf((float)N);
  orc_cf_t *tin = malloc(sizeof(orc_cf_t) * N), *tout = malloc(sizeof(orc_cf_t) * N);
  for (int s = 0; s < 2 * q->nof_symbols; s++) {
    int pos, cpl;
    sym_layout(q, s, &pos, &cpl);
    for (int n = 0; n < N; n++) {
      orc_cf_t v = in_time[pos + cpl + n];
      tin[n]     = q->freq_shift ? cmul(v, shift_val(q, cpl + n, cpl)) : v;
    }
    dft_any(tin, tout, N, 1, q->exact);
    orc_cf_t* o = &out_grid[s * nre];
    for (int i = 0; i < nre / 2; i++) {
      o[i]           = tout[N - nre / 2 + i];
      o[nre / 2 + i] = tout[dc + i];
    }
    if (q->normalize) {
      for (int i = 0; i < nre; i++) { o[i].re *= norm; o[i].im *= norm; }
    }
  }
  free(tin);
  free(tout);
}

void orc_ofdm_tx_sf(const orc_ofdm_t* q, const orc_cf_t* in_grid, orc_cf_t* out_time)
{ /* ofdm.c:488-530 (tx_slot, guru path) + :591-593 */
  int       N = q->symbol_sz, nre = q->nof_re, dc = q->freq_shift ? 0 : 1;
  float     norm = 1.0f / sqrtf((float)N);
  orc_cf_t *tin = malloc(sizeof(orc_cf_t) * N), *tout = malloc(sizeof(orc_cf_t) * N);
  for (int s = 0; s < 2 * q->nof_symbols; s++) {
    int pos, cpl;
    sym_layout(q, s, &pos, &cpl);
    memset(tin, 0, sizeof(orc_cf_t) * N);
    const orc_cf_t* g = &in_grid[s * nre];
    for (int i = 0; i < nre / 2; i++) {
      tin[dc + i]          = g[nre / 2 + i];
      tin[N - nre / 2 + i] = g[i];
    }
    dft_any(tin, tout, N, 0, q->exact);
    orc_cf_t* o = &out_time[pos];
    for (int n = 0; n < N; n++) {
      orc_cf_t v = tout[n];
      if (q->normalize) { v.re *= norm; v.im *= norm; }
      o[cpl + n] = v;
    }
    for (int n = 0; n < cpl; n++) {
      o[n] = o[N + n];
    }
    if (q->freq_shift) {
      for (int n = 0; n < cpl + N; n++) {
        o[n] = cmul(o[n], shift_val(q, n, cpl));
      }
    }
  }
  free(tin);
  free(tout);
}

/* ------------------------------------------------------------------ SC-FDMA transform precoding */

bool orc_dft_precoding_valid_prb(uint32_t nof_prb)
{ /* dft_precoding.c:88-98: 2^a 3^b 5^c */
  if (nof_prb == 0) return false;
  uint32_t n = nof_prb;
  while (n % 2 == 0) n /= 2;
  while (n % 3 == 0) n /= 3;
  while (n % 5 == 0) n /= 5;
  return n == 1;
}

int orc_dft_precoding(const orc_cf_t* in, orc_cf_t* out, uint32_t nof_prb, uint32_t nof_symbols, int forward, bool exact)
{ /* dft_precoding.c:100-113: nof_symbols independent N = 12*nof_prb DFTs, 1/sqrt(N) (plan norm, dft_fftw.c:294-297) */
  if (!orc_dft_precoding_valid_prb(nof_prb)) return -1;
  int   N    = 12 * (int)nof_prb;
  float norm = 1.0f / sqrtf((float)N);
  for (uint32_t s = 0; s < nof_symbols; s++) {
    dft_any(&in[s * N], &out[s * N], N, forward, exact);
    for (int i = 0; i < N; i++) { out[s * N + i].re *= norm; out[s * N + i].im *= norm; }
  }
  return 0;
}
