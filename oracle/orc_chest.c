/*
 * oracle/orc_chest.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 * CPU restatement of srslte_chest_dl_estimate_cfg for one rx antenna / one tx port (FDD, normal
 * subframes): LS pilot estimates -> noise from pilots -> Gauss/triangle smoothing (+ optional
 * time averaging) -> linear interpolation in frequency and time -> scalar measurements.
 * Follows chest_dl.c:304-379 (noise), :415-511 (interpolate), :513-556 (average), :558-569 (rssi),
 * :573-596 (cfo), :598-716 (per-port driver), :718-908 (aggregation); chest_common.c:62-88;
 * interp.c:145-168,240-267; convolution.c:180-218 (the "extrapolates extremes" variant is the one compiled).
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef orc_cf_t cf;
static inline cf c_add(cf a, cf b) { return (cf){a.re + b.re, a.im + b.im}; }
static inline cf c_sub(cf a, cf b) { return (cf){a.re - b.re, a.im - b.im}; }
static inline cf c_scale(cf a, float s) { return (cf){a.re * s, a.im * s}; }
static inline cf c_mulconj(cf a, cf b) { return (cf){a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; } /* a * conj(b) */

static float avg_power(const cf* x, uint32_t n)
{ /* vector.c:365-367 */
  float acc = 0;
  for (uint32_t i = 0; i < n; i++) acc += x[i].re * x[i].re + x[i].im * x[i].im;
  return acc / n;
}

static uint32_t gauss_filter(float* filter, uint32_t order, float std_dev)
{ /* chest_common.c:70-88 */
  uint32_t len = order + 1;
  int      center = (int)(len - 1) / 2;
  float    norm = 0;
  for (uint32_t i = 0; i < len; i++) {
    filter[i] = expf(-powf((float)((int)i - center), 2) / (2.0f * powf(std_dev, 2)));
    norm += filter[i];
  }
  for (uint32_t i = 0; i < len; i++) filter[i] *= 1.0f / norm;
  return len;
}

static void conv_same_cf(const cf* in, const float* h, cf* out, uint32_t N, uint32_t M)
{ /* convolution.c:180-218 */
  cf* first = malloc(sizeof(cf) * (M + M / 2));
  cf* last  = malloc(sizeof(cf) * (M + M / 2));
  for (uint32_t i = 0; i < M + M / 2; i++) {
    if (i < M / 2) {
      first[i] = c_sub(c_scale(in[1], (float)(2 + M / 2 - i)), c_scale(in[0], (float)(1 + M / 2 - i)));
    } else {
      first[i] = in[i - M / 2];
    }
    if (i >= M - 1) {
      last[i] = c_sub(c_scale(in[N - 1], (float)(2 + i - M / 2)), c_scale(in[N - 2], (float)(1 + i - M / 2)));
    } else {
      last[i] = in[N - M + i + 1];
    }
  }
  uint32_t i = 0, j = 0;
  for (; i < N; i++) {
    const cf* src = i < M / 2 ? &first[i] : (i < N - M / 2 ? &in[i - M / 2] : &last[j++]);
    cf acc = {0, 0};
    for (uint32_t t = 0; t < M; t++) acc = c_add(acc, c_scale(src[t], h[t]));
    out[i] = acc;
  }
  free(first);
  free(last);
}

static void interp_linear_offset(const cf* in, cf* out, uint32_t L, uint32_t M, uint32_t off_st, uint32_t off_end)
{ /* interp.c:240-267 */
  for (uint32_t j = 0; j < off_st; j++) {
    cf d = c_sub(in[1], in[0]);
    out[off_st - j - 1] = c_sub(in[0], c_scale(c_scale(d, (float)(j + 1)), 1.0f / M));
  }
  for (uint32_t i = 0; i + 1 < L; i++) {
    cf d = c_scale(c_sub(in[i + 1], in[i]), 1.0f / (float)M);
    for (uint32_t j = 0; j < M; j++) out[i * M + j + off_st] = c_add(in[i], c_scale(d, (float)j));
  }
  if (L > 1) {
    cf d = c_sub(in[L - 1], in[L - 2]);
    for (uint32_t j = 0; j < off_end; j++) out[(L - 1) * M + j + off_st] = c_add(in[L - 1], c_scale(c_scale(d, (float)j), 1.0f / M));
  }
}

static void interp_vector(const cf* in0, const cf* in1, const cf* start, cf* between, uint32_t dist, uint32_t M, uint32_t len)
{ /* interp.c:145-168 (to_right) */
  cf* diff = malloc(sizeof(cf) * len);
  for (uint32_t i = 0; i < len; i++) diff[i] = c_scale(c_sub(in1[i], in0[i]), (float)1 / dist);
  const cf* s = start ? start : in0;
  for (uint32_t i = 0; i < len; i++) between[i] = c_add(s[i], diff[i]);
  for (uint32_t m = 0; m + 1 < M; m++) {
    for (uint32_t i = 0; i < len; i++) between[len + i] = c_add(between[i], diff[i]);
    between += len;
  }
  free(diff);
}

/* estimate_noise_pilots (chest_dl.c:304-379): rows[nsym][nref] pilot estimates; tmp holds 3 (nref + 2) values */
static float noise_pilots(cf* est, uint32_t nref, uint32_t nsym, uint32_t fidx0, cf* tmp)
{
  if (nsym == 1) { /* "Special case for 1 symbol" (chest_dl.c:322-331): residual against the mean of the two neighbours and the pilot itself */
    for (uint32_t k = 0; k + 2 < nref; k++) {
      cf t = c_add(c_add(est[k + 1], est[k]), est[k + 2]);
      tmp[k] = c_sub(est[k + 1], c_scale(t, 1.0f / 3.0f));
    }
    return avg_power(tmp, nref - 2);
  }
  cf* in2d[6];
  for (uint32_t i = 0; i < nsym; i++) in2d[i + 1] = &est[i * nref];
  in2d[0]        = &tmp[nref];
  in2d[nsym + 1] = &tmp[2 * nref];
  for (uint32_t k = 0; k < nref; k++) {
    if (nsym > 3) { /* virtual rows before the first and after the last pilot symbol: linear extrapolation (:337-350) */
      in2d[0][k]        = c_sub(c_scale(in2d[2][k], 2.0f), in2d[4][k]);
      in2d[nsym + 1][k] = c_sub(c_scale(in2d[nsym - 1][k], 2.0f), in2d[nsym - 3][k]);
    } else { /* two or three symbols: copies of the second / the last but one row */
      in2d[0][k]        = in2d[2][k];
      in2d[nsym + 1][k] = in2d[nsym - 1][k];
    }
  }
  float sum_power = 0;
  int   count     = 0;
  for (uint32_t i = 1; i < nsym + 1; i++) {
    uint32_t off = ((fidx0 < 3) ^ (i & 1)) ? 0 : 1;
    for (uint32_t k = 0; k < nref; k++) tmp[k] = in2d[i][k];
    for (int side = -1; side <= 1; side += 2) {
      const cf* nb = in2d[(int)i + side];
      for (uint32_t k = 0; k < nref - off; k++) tmp[off + k] = c_add(tmp[off + k], nb[k]);
      for (uint32_t k = 0; k < nref + off - 1; k++) tmp[k] = c_add(tmp[k], nb[1 - off + k]);
      if (off) {
        tmp[0] = c_add(tmp[0], c_sub(c_scale(nb[0], 2.0f), nb[1]));
      } else {
        tmp[nref - 1] = c_add(tmp[nref - 1], c_sub(c_scale(nb[nref - 2], 2.0f), nb[nref - 1]));
      }
    }
    for (uint32_t k = 0; k < nref; k++) tmp[k] = c_sub(in2d[i][k], c_scale(tmp[k], 1.0f / 5.0f));
    sum_power = avg_power(tmp, nref); /* '=' not '+=': upstream quirk (chest_dl.c:374) */
    count++;
  }
  return sum_power / (float)count * sqrtf(5.0f);
}

/* estimate_port for one port of one receive antenna (chest_dl.c:598-716); raw = {noise, rsrp, rssi, cfo, sync, corr} of that
   (antenna, port). Ports 0/1 have 4 pilot symbols per subframe, ports 2/3 two (symbols 1 and 8). est is the estimator's
   q->pilot_estimates, [4][2 nof_prb], SHARED by the ports of an antenna as upstream: chest_estimate_cfo (:573-596) always pairs its
   first and second half, so for ports 2/3 it multiplies their two symbols with what port 1 left in the second half.
   interpolate_subframe with ports 2/3: their nsymbols is 2 < 3, so upstream takes the copy branch (:467-471) and replicates symbol 0 of ce
   over the subframe - a symbol this call did not write (the two interpolated rows, symbols 1 and 8, are overwritten): ce is in / out. */
void orc_pss_generate(uint32_t N_id_2, orc_cf_t* signal /* [62] */)
{ /* srslte_pss_generate (pss.c:348-376): Zadoff-Chu roots 25, 29, 34 with the DC element left out */
  const float root_value[] = {25.0, 29.0, 34.0};
  for (int i = 0; i < 62; i++) {
    float arg = i < 31 ? (float)-1 * M_PI * root_value[N_id_2 % 3] * ((float)i * ((float)i + 1.0)) / 63.0
                       : (float)-1 * M_PI * root_value[N_id_2 % 3] * (((float)i + 2.0) * ((float)i + 1.0)) / 63.0;
    signal[i].re = cosf(arg);
    signal[i].im = sinf(arg);
  }
}

static int chest_port(const orc_cell_t* cell, uint32_t sf_idx, const orc_chest_cfg_t* cfg, const orc_cf_t* grid, orc_cf_t* ce, uint32_t port,
                      cf* est, float raw[6], float noise_prev)
{
  /* pilot symbols of this port in this subframe: 4 / 2, fewer in a TDD special subframe (refsignal_dl.c:162-225) */
  const uint32_t P = cell->nof_prb, nre = 12 * P, nsym = orc_crs_nof_symbols(cell, sf_idx, port), nref = 2 * P, npil = nsym * nref;
  const uint32_t nsymb_sf = cell->cp_norm ? 14 : 12;
  if (port > 3) return -1;
  /* what upstream computes from rows the shortened subframe does not have: chest_estimate_cfo pairs rows (0, 2) and (1, 3) of a NORMAL subframe
     (:577-590), the extended-CP time interpolation has no special-subframe branch ("TODO", :497-502) */
  if (nsym < (port < 2 ? 4u : 2u) && (cfg->cfo_estimate_enable || (!cell->cp_norm && cfg->interpolate_subframe && nsym >= 3))) return -3;
  const bool stale = port > 1 && ce && cfg->interpolate_subframe;
  cf* known = malloc(sizeof(cf) * 4 * nref);
  cf* recv  = malloc(sizeof(cf) * 4 * nref);
  cf* avg   = malloc(sizeof(cf) * 4 * nref);
  cf* tmp   = malloc(sizeof(cf) * 3 * (nref + 2));
  orc_crs_pilots(cell, sf_idx, port, known);

  /* pilots + LS (chest_dl.c:684-690) */
  for (uint32_t l = 0; l < nsym; l++) {
    uint32_t sym = orc_crs_nsymbol(l, cell->cp_norm, port), fidx = orc_crs_fidx(cell, l, port);
    for (uint32_t i = 0; i < nref; i++) {
      recv[l * nref + i] = grid[sym * nre + fidx + 6 * i];
      est[l * nref + i]  = c_mulconj(recv[l * nref + i], known[l * nref + i]);
    }
  }
  float rsrp = avg_power(recv, npil); /* chest_dl.c:710 */
  float rssi = 0;                     /* chest_dl.c:558-569 */
  for (uint32_t l = 0; l < nsym; l++) {
    rssi += avg_power(&grid[orc_crs_nsymbol(l, cell->cp_norm, port) * nre], nre) * nre;
  }
  rssi /= nsym;

  float sync = NAN; /* chest_dl.c:692-703 with srslte_vec_estimate_frequency (vector_simd.c:1606-1656; its SIMD body multiplies by an approximate reciprocal) */
  if (cfg->sync_error_enable) {
    float k = (float)orc_symbol_sz((int)P) / 6.0f, sum = 0.0f;
    for (uint32_t l = 0; l < nsym; l++) {
      const cf* x = &est[l * nref];
      float     ss = 0.0f;
      for (uint32_t i = 1; i < nref; i++) {
        float pw = sqrtf((x[i].re * x[i].re + x[i].im * x[i].im) * (x[i - 1].re * x[i - 1].re + x[i - 1].im * x[i - 1].im));
        ss += (x[i].re * x[i - 1].im - x[i - 1].re * x[i].im) / pw;
      }
      sum += asinf(ss / (float)(nref - 1)) / (2.0f * (float)M_PI) * k;
    }
    sync = sum / nsym;
  }
  float corr = 0; /* chest_dl.c:706-709 */
  if (cfg->rsrp_neighbour) {
    cf acc = {0, 0};
    for (uint32_t i = 0; i < npil; i++) acc = c_add(acc, est[i]);
    double energy = sqrt((double)(acc.re / npil) * (acc.re / npil) + (double)(acc.im / npil) * (acc.im / npil));
    corr = (float)(energy * energy);
  }
  float cfo = 0;
  if (cfg->cfo_estimate_enable) { /* chest_dl.c:573-596 */
    float n = (float)orc_symbol_sz((int)P), ns = cell->cp_norm ? 7.0f : 6.0f, ng = (float)orc_cp_len_norm(1, (int)n); /* the normal-CP length whatever the cell's (:577) */
    cf    sum = {0, 0};
    for (uint32_t i = 0; i < 2; i++) { /* npilots is port 0's whatever the port (:582) */
      for (uint32_t k = 0; k < nref; k++) sum = c_add(sum, c_mulconj(est[i * nref + k], est[(i + 2) * nref + k]));
    }
    cfo = (float)(-atan2f(sum.im, sum.re) * n / (ns * (n + ng)) / 2 / M_PI);
  }

  /* noise from pilots (chest_dl.c:304-379) */
  /* PSS / EMPTY: the estimate of the last subframe 0 or 5 stays (and sets the automatic Gauss filter) until ce is there (:657-672) */
  float noise = cfg->noise_alg == 0 ? noise_pilots(est, nref, nsym, orc_crs_fidx(cell, 0, port), tmp) : noise_prev;

  if (ce) {
    float    filter[64];
    uint32_t flen = 0;
    if (cfg->filter_type == 0) { /* chest_dl.c:628-637 */
      flen = cfg->filter_coef[0] <= 0 ? gauss_filter(filter, 4, noise * 200.0f) : gauss_filter(filter, (uint32_t)cfg->filter_coef[0], cfg->filter_coef[1]);
    } else if (cfg->filter_type == 1) { /* chest_common.c:62-68 */
      filter[0] = cfg->filter_coef[0]; filter[2] = cfg->filter_coef[0]; filter[1] = 1 - 2 * cfg->filter_coef[0];
      flen = 3;
    }
    const cf* pil = est;
    if (cfg->filter_type != 2) { /* average_pilots, chest_dl.c:513-556 */
      uint32_t n = nref, ns = nsym;
      if (!cfg->interpolate_subframe && nsym > 1) { /* :527-545; with three rows only the first two are summed, yet scaled by 2 / 3 */
        bool first_low = orc_crs_fidx(cell, 0, port) < 3;
        for (uint32_t k = 0; k < nref; k++) {
          cf a = est[k], b = est[nref + k];
          if (nsym == 4) {
            a = c_add(a, est[2 * nref + k]);
            b = c_add(b, est[3 * nref + k]);
          }
          avg[2 * k]     = first_low ? a : b;
          avg[2 * k + 1] = first_low ? b : a;
        }
        n = 2 * nref;
        for (uint32_t k = 0; k < n; k++) est[k] = c_scale(avg[k], 2.0f / (float)nsym);
        ns = 1;
      }
      for (uint32_t l = 0; l < ns; l++) conv_same_cf(&est[l * n], filter, &avg[l * n], n, flen);
      pil = avg;
    }
    /* interpolate_pilots, chest_dl.c:415-511 */
    if (stale) {
      for (uint32_t l = 1; l < nsymb_sf; l++) memcpy(&ce[l * nre], ce, sizeof(cf) * nre);
    } else if (!cfg->interpolate_subframe && nsym > 1) {
      uint32_t off = cell->id % 3;
      interp_linear_offset(pil, ce, 4 * P, 3, off, 3 - off);
      for (uint32_t l = 1; l < nsymb_sf; l++) memcpy(&ce[l * nre], ce, sizeof(cf) * nre);
    } else if (!cfg->interpolate_subframe || nsym < 3) { /* one pilot symbol, or two with interpolate_subframe (:433,:456-471): the rows, then symbol 0 everywhere */
      for (uint32_t l = 0; l < (cfg->interpolate_subframe ? nsym : 1); l++) {
        uint32_t off = orc_crs_fidx(cell, l, port);
        interp_linear_offset(&pil[nref * l], &ce[orc_crs_nsymbol(l, cell->cp_norm, port) * nre], nref, 6, off, 6 - off);
      }
      for (uint32_t l = 1; l < nsymb_sf; l++) memcpy(&ce[l * nre], ce, sizeof(cf) * nre);
    } else {
      for (uint32_t l = 0; l < nsym; l++) {
        uint32_t off = orc_crs_fidx(cell, l, port);
        interp_linear_offset(&pil[nref * l], &ce[orc_crs_nsymbol(l, cell->cp_norm, port) * nre], nref, 6, off, 6 - off);
      }
#define S(i) (&ce[(i) * nre])
      if (cell->cp_norm && nsym == 3) { /* :481-488: symbols 8-13 continue the 4 -> 7 slope */
        interp_vector(S(0), S(4), NULL, S(1), 4, 3, nre);
        interp_vector(S(4), S(7), NULL, S(5), 3, 2, nre);
        interp_vector(S(4), S(7), S(7), S(8), 3, 6, nre);
      } else if (cell->cp_norm) {
        interp_vector(S(0), S(4), NULL, S(1), 4, 3, nre);
        interp_vector(S(4), S(7), NULL, S(5), 3, 2, nre);
        interp_vector(S(7), S(11), NULL, S(8), 4, 3, nre);
        interp_vector(S(7), S(11), S(11), S(12), 4, 2, nre);
      } else { /* extended CP: pilot symbols 0, 3, 6, 9 of 12 (chest_dl.c:497-502) */
        interp_vector(S(0), S(3), NULL, S(1), 3, 2, nre);
        interp_vector(S(3), S(6), NULL, S(4), 3, 2, nre);
        interp_vector(S(6), S(9), NULL, S(7), 3, 2, nre);
        interp_vector(S(6), S(9), S(9), S(10), 3, 2, nre);
      }
#undef S
    }
    if (cfg->noise_alg != 0 && (sf_idx == 0 || sf_idx == 5)) {
      const uint32_t nsl = cell->cp_norm ? 7 : 6, k_pss = (nsl - 1) * nre + nre / 2 - 31, k_sss = (nsl - 2) * nre + nre / 2 - 31;
      if (cfg->noise_alg == 1) { /* estimate_noise_pss (chest_dl.c:381-398): |ce pss - received|^2 over the 62 PSS carriers */
        cf pss[62], d[62];
        orc_pss_generate(cell->id % 3, pss);
        for (int i = 0; i < 62; i++) {
          cf h = ce[k_pss + i], x = pss[i];
          d[i] = c_sub((cf){h.re * x.re - h.im * x.im, h.re * x.im + h.im * x.re}, grid[k_pss + i]);
        }
        noise = (float)(cell->nof_ports * avg_power(d, 62) / sqrt(2));
      } else { /* estimate_noise_empty_sc (:401-411): the 5 empty carriers either side of SSS and PSS */
        noise = 0;
        noise += avg_power(&grid[k_sss - 5], 5);
        noise += avg_power(&grid[k_sss + 62], 5);
        noise += avg_power(&grid[k_pss - 5], 5);
        noise += avg_power(&grid[k_pss + 62], 5);
      }
    }
  }

  raw[0] = noise; raw[1] = rsrp; raw[2] = rssi; raw[3] = cfo; raw[4] = sync; raw[5] = corr;
  free(known); free(recv); free(avg); free(tmp);
  return 0;
}

static void fill_res(uint32_t P, uint32_t nof_rx, uint32_t nof_ports, float raw[4][4][6] /* [antenna][port] */, bool cfo_enable, orc_chest_res_t* res)
{ /* chest_dl.c:747-871: noise is averaged over ports and antennas; RSSI and RSRQ use port 0 and are averaged over the antennas;
     get_rsrp (:809-819) takes the maximum over "ports" indexed by the ANTENNA counter of the antenna-mean RSRP of that port (0 for a
     port that was never estimated); q->cfo is overwritten by every (antenna, port) estimate in turn, so the last one survives */
  float noise = 0, rssi = 0, rsrq = 0;
  for (uint32_t a = 0; a < nof_rx; a++) {
    float n = 0;
    for (uint32_t p = 0; p < nof_ports; p++) n += raw[a][p][0];
    noise += n / nof_ports;
    rssi += 4 * raw[a][0][2] / P / 12;
    rsrq += P * raw[a][0][1] / raw[a][0][2];
  }
  noise /= nof_rx; rssi /= nof_rx; rsrq /= nof_rx;
  float rsrp = -1e9f;
  for (uint32_t i = 0; i < nof_rx; i++) {
    float v = 0;
    if (i < nof_ports) {
      for (uint32_t a = 0; a < nof_rx; a++) v += raw[a][i][1];
      v /= nof_rx;
    }
    if (v > rsrp) rsrp = v;
  }
  memset(res, 0, sizeof(*res));
  res->noise_estimate     = noise;
  res->noise_estimate_dbm = (float)(10 * log10(noise) + 30);
  res->cfo                = cfo_enable ? raw[nof_rx - 1][nof_ports - 1][3] : 0.f;
  res->rsrp               = rsrp;
  res->rsrp_dbm           = (float)(10 * log10(rsrp) + 30);
  res->rsrq               = rsrq;
  res->rsrq_db            = (float)(10 * log10(rsrq));
  res->snr_db             = (float)(10 * log10(rsrp / noise));
  res->rssi_dbm           = (float)(10 * log10(rssi) + 30);
  res->sync_error         = raw[0][0][4]; /* "Take only the channel used for synch" (chest_dl.c:859) */
  float neigh = -1e9f;                    /* get_rsrp_neighbour (chest_dl.c:821-843): max over antennas of the port-mean correlation power */
  for (uint32_t i = 0; i < nof_rx; i++) {
    float v = 0;
    for (uint32_t j = 0; j < nof_ports; j++) v += raw[i][j][5];
    v /= nof_ports;
    if (v > neigh) neigh = v;
  }
  res->rsrp_neigh = neigh;
}

int orc_chest_dl_ports(const orc_cell_t* cell, uint32_t sf_idx, const orc_chest_cfg_t* cfg, uint32_t nof_rx, const orc_cf_t* const* grid,
                       orc_cf_t* const* ce /* [port * nof_rx + antenna] */, orc_chest_res_t* res, float* raw_out /* [antenna][port][4] or NULL */)
{
  return orc_chest_dl_ports_state(cell, sf_idx, cfg, nof_rx, grid, ce, res, raw_out, NULL);
}

int orc_chest_dl_ports_state(const orc_cell_t* cell, uint32_t sf_idx, const orc_chest_cfg_t* cfg, uint32_t nof_rx, const orc_cf_t* const* grid,
                             orc_cf_t* const* ce, orc_chest_res_t* res, float* raw_out, float* noise_state /* [antenna][port] in/out or NULL */)
{ /* srslte_chest_dl_estimate_cfg (chest_dl.c:884-908) for cell->nof_ports in {1, 2, 4} and nof_rx receive antennas; noise_state is the
     estimator's q->noise_estimate, which the PSS / EMPTY algorithms only renew in subframes 0 and 5 */
  float raw[4][4][6];
  memset(raw, 0, sizeof(raw));
  if (nof_rx < 1 || nof_rx > 4 || (cell->nof_ports != 1 && cell->nof_ports != 2 && cell->nof_ports != 4)) return -1;
  cf* est = calloc(8 * cell->nof_prb, sizeof(cf)); /* q->pilot_estimates */
  for (uint32_t a = 0; a < nof_rx; a++) {
    for (uint32_t p = 0; p < cell->nof_ports; p++) {
      int r = chest_port(cell, sf_idx, cfg, grid[a], ce ? ce[p * nof_rx + a] : NULL, p, est, raw[a][p],
                         noise_state ? noise_state[a * cell->nof_ports + p] : 0.0f);
      if (r) {
        free(est);
        return r;
      }
      if (noise_state) noise_state[a * cell->nof_ports + p] = raw[a][p][0];
    }
  }
  free(est);
  if (res) fill_res(cell->nof_prb, nof_rx, cell->nof_ports, raw, cfg->cfo_estimate_enable, res);
  if (raw_out) {
    for (uint32_t a = 0; a < nof_rx; a++) {
      for (uint32_t p = 0; p < cell->nof_ports; p++) memcpy(&raw_out[(a * cell->nof_ports + p) * 4], raw[a][p], sizeof(float) * 4);
    }
  }
  return 0;
}

int orc_chest_dl(const orc_cell_t* cell, uint32_t sf_idx, const orc_chest_cfg_t* cfg, const orc_cf_t* grid, orc_cf_t* ce, orc_chest_res_t* res)
{ /* one antenna, one port */
  orc_cell_t c1 = *cell;
  c1.nof_ports  = 1;
  return orc_chest_dl_ports(&c1, sf_idx, cfg, 1, &grid, ce ? &ce : NULL, res, NULL);
}

int orc_chest_dl_multi(const orc_cell_t* cell, uint32_t sf_idx, const orc_chest_cfg_t* cfg, uint32_t nof_rx, const orc_cf_t* const* grid,
                       orc_cf_t* const* ce, orc_chest_res_t* res)
{ /* nof_rx receive antennas, one port */
  orc_cell_t c1 = *cell;
  c1.nof_ports  = 1;
  return orc_chest_dl_ports(&c1, sf_idx, cfg, nof_rx, grid, ce, res, NULL);
}

/* ------------------------------------------------------------------ MBSFN subframes: the reference signal of antenna
 * port 4 (refsignal_dl.c:328-400) and estimate_port_mbsfn (chest_dl.c:718-745) with the MBSFN branches of average_pilots (:513-556),
 * interpolate_pilots (:415-511) and estimate_noise_pilots (:304-379). The subframe has 12 symbols (extended CP); symbol 0 carries the
 * cell's CRS, symbols 2, 6, 10 a pilot on every second sub-carrier (offsets 0, 1, 0). */
static const uint32_t MBSFN_SYM[3] = {2, 6, 10}, MBSFN_FIDX[3] = {0, 1, 0};

int orc_mbsfn_pilots(uint32_t nof_prb, uint32_t area_id, uint32_t sf_idx, orc_cf_t* pilots /* [3][6 nof_prb] */)
{ /* refsignal_dl.c:361-400: c_init = 512 (7 (ns + 1) + l' + 1)(2 N_mbsfn + 1) + N_mbsfn, offset 3 (MAX_PRB - nof_prb) */
  const uint32_t MAX_PRB = 110;
  uint8_t*       c = malloc(20 * MAX_PRB);
  if (nof_prb < 6 || nof_prb > MAX_PRB || area_id > 255 || sf_idx > 9) {
    free(c);
    return -1;
  }
  for (uint32_t l = 0; l < 3; l++) {
    uint32_t lp = MBSFN_SYM[l] % 6, slot = l ? 2 * sf_idx + 1 : 2 * sf_idx;
    orc_gold(512 * (7 * (slot + 1) + lp + 1) * (2 * area_id + 1) + area_id, 20 * MAX_PRB, c);
    for (uint32_t i = 0; i < 6 * nof_prb; i++) {
      uint32_t mp = i + 3 * (MAX_PRB - nof_prb);
      pilots[6 * nof_prb * l + i].re = (float)((1 - 2 * (float)c[2 * mp]) / sqrt(2));
      pilots[6 * nof_prb * l + i].im = (float)((1 - 2 * (float)c[2 * mp + 1]) / sqrt(2));
    }
  }
  free(c);
  return 0;
}

int orc_mbsfn_put_sf(const orc_cell_t* cell, uint32_t sf_idx, uint32_t port_id, uint32_t area_id, orc_cf_t* grid)
{ /* srslte_refsignal_mbsfn_put_sf (refsignal_dl.c:297-326): the first CRS symbol of the port + the three MBSFN pilot symbols */
  const uint32_t P = cell->nof_prb, nre = 12 * P;
  cf*            crs = malloc(sizeof(cf) * 8 * P);
  cf*            mb  = malloc(sizeof(cf) * 18 * P);
  if (port_id > 3 || orc_mbsfn_pilots(P, area_id, sf_idx, mb)) {
    free(crs); free(mb);
    return -1;
  }
  orc_crs_pilots(cell, sf_idx, port_id, crs);
  uint32_t fidx = orc_crs_fidx(cell, 0, port_id);
  for (uint32_t i = 0; i < 2 * P; i++) grid[fidx + 6 * i] = crs[i]; /* always symbol 0 (:308) */
  for (uint32_t l = 0; l < 3; l++) {
    for (uint32_t i = 0; i < 6 * P; i++) grid[MBSFN_SYM[l] * nre + MBSFN_FIDX[l] + 2 * i] = mb[6 * P * l + i];
  }
  free(crs); free(mb);
  return 0;
}

int orc_chest_dl_mbsfn(const orc_cell_t* cell, uint32_t sf_idx, const orc_chest_cfg_t* cfg, uint32_t area_id, uint32_t port,
                       const orc_cf_t* grid, orc_cf_t* ce /* 12 symbols written */, float* noise_out)
{ /* estimate_port_mbsfn (chest_dl.c:718-745) + chest_interpolate_noise_est (:598-676) in MBSFN mode. The reference leaves rsrp, rssi,
     cfo and the sync error of the (antenna, port) as the last normal subframe set them; with the PSS / EMPTY noise algorithms the noise
     estimate stays too (MBSFN subframes are never 0 or 5): *noise_out is NaN then. Without interpolate_subframe the reference
     estimates symbol 0 only and interpolates from never-written symbols (-3 here), as it does for ports 2/3 (symbol 0 unwritten). */
  const uint32_t P = cell->nof_prb, nre = 12 * P, ncrs = 2 * P, nmb = 6 * P, npil = 20 * P;
  if (port > 1 || !cfg->interpolate_subframe) return -3;
  cf* known = malloc(sizeof(cf) * 8 * P);
  cf* mb    = malloc(sizeof(cf) * 18 * P);
  cf* est   = malloc(sizeof(cf) * npil);
  cf* avg   = malloc(sizeof(cf) * npil);
  cf* tmp   = malloc(sizeof(cf) * 3 * (npil / 3 + 2));
  if (orc_mbsfn_pilots(P, area_id, sf_idx, mb)) {
    free(known); free(mb); free(est); free(avg); free(tmp);
    return -1;
  }
  orc_crs_pilots(cell, sf_idx, port, known);
  uint32_t fidx = orc_crs_fidx(cell, 0, port); /* srslte_refsignal_mbsfn_get_sf (refsignal_dl.c:455-487) + LS (:734-741) */
  for (uint32_t i = 0; i < ncrs; i++) est[i] = c_mulconj(grid[orc_crs_nsymbol(0, cell->cp_norm, port) * nre + fidx + 6 * i], known[i]);
  for (uint32_t l = 0; l < 3; l++) {
    for (uint32_t i = 0; i < nmb; i++) est[ncrs + nmb * l + i] = c_mulconj(grid[MBSFN_SYM[l] * nre + MBSFN_FIDX[l] + 2 * i], mb[nmb * l + i]);
  }
  /* REFS noise: estimate_noise_pilots walks the 20 nof_prb estimates as 3 rows of 20 nof_prb / 3 (integer), CRS row included (:310-315) */
  float noise = cfg->noise_alg == 0 ? noise_pilots(est, npil / 3, 3, MBSFN_FIDX[1], tmp) : NAN;
  if (noise_out) *noise_out = noise;
  if (ce) {
    float    filter[64];
    uint32_t flen = 0;
    if (cfg->filter_type == 0) { /* the reference warns that Gauss is "not supported" and applies it (:628-637) */
      flen = cfg->filter_coef[0] <= 0 ? gauss_filter(filter, 4, noise * 200.0f) : gauss_filter(filter, (uint32_t)cfg->filter_coef[0], cfg->filter_coef[1]);
    } else if (cfg->filter_type == 1) {
      filter[0] = cfg->filter_coef[0]; filter[2] = cfg->filter_coef[0]; filter[1] = 1 - 2 * cfg->filter_coef[0];
      flen = 3;
    }
    const cf* pil = est;
    if (cfg->filter_type != 2) { /* average_pilots: the CRS row is copied, each MBSFN row smoothed (:546-555) */
      memcpy(avg, est, sizeof(cf) * ncrs);
      for (uint32_t l = 0; l < 3; l++) conv_same_cf(&est[ncrs + nmb * l], filter, &avg[ncrs + nmb * l], nmb, flen);
      pil = avg;
    }
    interp_linear_offset(pil, &ce[orc_crs_nsymbol(0, cell->cp_norm, port) * nre], ncrs, 6, fidx, 6 - fidx); /* :436-440 */
    for (uint32_t l = 0; l < 3; l++) {                                                                      /* :441-447 */
      interp_linear_offset(&pil[ncrs + nmb * l], &ce[MBSFN_SYM[l] * nre], nmb, 2, MBSFN_FIDX[l], MBSFN_FIDX[l] ? 1 : 2);
    }
#define S(i) (&ce[(i) * nre])
    interp_vector(S(0), S(2), NULL, S(1), 2, 1, nre); /* :474-478 */
    interp_vector(S(2), S(6), NULL, S(3), 4, 3, nre);
    interp_vector(S(6), S(10), NULL, S(7), 4, 3, nre);
    interp_vector(S(6), S(10), S(10), S(11), 4, 1, nre);
#undef S
  }
  free(known); free(mb); free(est); free(avg); free(tmp);
  return 0;
}

/* ------------------------------------------------------------------ UL: PUSCH DMRS (refsignal_ul.c) and srslte_chest_ul_estimate_pusch
 * (chest_ul.c), SURVEY §8f N3. Grants of 1 and 2 PRB use the tabulated QPSK base sequences of 36.211 Tables 5.5.1.2-1/-2; from 3 PRB
 * on the base sequence is a Zadoff-Chu extension. */

static const uint32_t N_DMRS_1[8] = {0, 2, 3, 4, 6, 8, 9, 10}; /* 36.211 Table 5.5.2.1.1-2 (refsignal_ul.c:42) */
static const uint32_t N_DMRS_2[8] = {0, 6, 3, 4, 2, 8, 10, 9}; /* 36.211 Table 5.5.2.1.1-1 (refsignal_ul.c:39) */

int orc_ul_dmrs_init(orc_ul_dmrs_t* q, uint32_t cell_id) { return orc_ul_dmrs_init_cp(q, cell_id, 7); }

int orc_ul_dmrs_init_cp(orc_ul_dmrs_t* q, uint32_t cell_id, uint32_t nsl)
{ /* srslte_refsignal_ul_set_cell (refsignal_ul.c:206-238); nsl = SRSLTE_CP_NSYMB(cell.cp): the cyclic-shift hopping n_PRS reads 8 bits per
     SC-FDMA symbol of the slot, so the CP sets the stride (:127-133) */
  memset(q, 0, sizeof(*q));
  q->cell_id = cell_id;
  uint8_t c[8 * 7 * 20];
  if (nsl != 6 && nsl != 7) return -1;
  for (uint32_t ds = 0; ds < 30; ds++) { /* generate_n_prs :118-141 and generate_srslte_sequence_hopping_v :149-163 share the seed */
    uint32_t c_init = ((cell_id / 30) << 5) + (((cell_id % 30) + ds) % 30);
    orc_gold(c_init, 8 * nsl * 20, c);
    for (uint32_t ns = 0; ns < 20; ns++) {
      uint32_t n = 0;
      for (int i = 0; i < 8; i++) n += (uint32_t)c[8 * nsl * ns + i] << i;
      q->n_prs[ds][ns] = n;
      q->v[ns][ds]     = c[ns];
    }
  }
  orc_gold(cell_id / 30, 160, c); /* srslte_group_hopping_f_gh, phy_common.c:419-436 */
  for (uint32_t ns = 0; ns < 20; ns++) {
    q->f_gh[ns] = 0;
    for (int i = 0; i < 8; i++) q->f_gh[ns] += (uint32_t)c[8 * ns + i] << i;
  }
  return 0;
}

static uint32_t largest_prime_below(uint32_t x)
{ /* refsignal_ul.c:240-249 looks the value up in a table of primes */
  for (uint32_t p = x - 1; p >= 2; p--) {
    int prime = 1;
    for (uint32_t d = 2; d * d <= p; d++) {
      if (p % d == 0) { prime = 0; break; }
    }
    if (prime) return p;
  }
  return 0;
}

/* 36.211 Tables 5.5.1.2-1 and 5.5.1.2-2: phi(n) of the QPSK base sequences for M_sc = 12 and 24 (one and two PRB), one string per
 * group u; digit d stands for phi = 2 d - 3, i.e. 0 1 2 3 = -3 -1 +1 +3 (the reference keeps them as int arrays, ul_rs_tables.h) */
static const char* const PHI_12[30] = {
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
static const char* const PHI_24[30] = {
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

int orc_ul_dmrs_pusch_gen(const orc_ul_dmrs_t* q, const orc_ul_dmrs_cfg_t* cfg, uint32_t nof_prb, uint32_t sf_idx, uint32_t n_dmrs, orc_cf_t* r)
{ /* srslte_refsignal_dmrs_pusch_gen (refsignal_ul.c:459-487) with compute_r :352-371, arg_r_uv_mprb :271-283, get_q :257-269,
     pusch_alpha :296-304. The float/double mix of the reference is kept operation by operation: at 100 PRB the argument reaches
     4e6 rad, where a float has a resolution of 0.5 rad, so every rounding is visible in the sequence. */
  if (cfg->cyclic_shift >= 8 || cfg->delta_ss >= 30 || n_dmrs >= 8 || sf_idx >= 10) return -1;
  const uint32_t M_sc = 12 * nof_prb, N_sz = nof_prb >= 3 ? largest_prime_below(M_sc) : 1;
  for (uint32_t ns = 2 * sf_idx; ns < 2 * (sf_idx + 1); ns++) {
    uint32_t u = ((cfg->group_hopping_en ? q->f_gh[ns] : 0) + (q->cell_id % 30) + cfg->delta_ss) % 30;
    uint32_t v = (nof_prb >= 6 && cfg->sequence_hopping_en) ? q->v[ns][cfg->delta_ss] : 0;
    float    n_sz = (float)N_sz, q_hat = n_sz * (u + 1) / 31, qf;
    if ((((uint32_t)(2 * q_hat)) % 2) == 0) {
      qf = (float)(q_hat + 0.5 + v);
    } else {
      qf = (float)(q_hat + 0.5 - v);
    }
    float    qq    = (float)(uint32_t)qf;
    uint32_t n_cs  = (N_DMRS_1[cfg->cyclic_shift] + N_DMRS_2[n_dmrs] + q->n_prs[cfg->delta_ss][ns]) % 12;
    float    alpha = (float)(2 * M_PI * n_cs / 12);
    for (uint32_t i = 0; i < M_sc; i++) {
      float m   = (float)(i % N_sz);
      float arg = (float)(-M_PI * qq * m * (m + 1) / n_sz);
      if (nof_prb < 3) { /* one and two PRB: tabulated QPSK sequences, arg = phi(n) pi / 4 (refsignal_ul.c:143-147,:251-255) */
        int phi = 2 * ((nof_prb == 1 ? PHI_12 : PHI_24)[u][i] - '0') - 3;
        arg     = (float)(phi * M_PI / 4);
      }
#ifdef ORC_DMRS_NO_FMA
      float x = arg + alpha * (float)i;
#else
      float x = fmaf(alpha, (float)i, arg); /* the reference's -Ofast -mfma build contracts tmp_arg[i] + alpha*i */
#endif
      r[(ns % 2) * M_sc + i] = (orc_cf_t){cosf(x), sinf(x)};
    }
  }
  return 0;
}

int orc_chest_ul_pusch(const orc_cf_t* r_dmrs, uint32_t cell_nof_prb, uint32_t L_prb, uint32_t n_prb, const orc_cf_t* grid, orc_cf_t* ce,
                       orc_chest_ul_res_t* res)
{
  return orc_chest_ul_pusch_hop(r_dmrs, cell_nof_prb, L_prb, n_prb, n_prb, grid, ce, res);
}

int orc_chest_ul_pusch_hop(const orc_cf_t* r_dmrs, uint32_t cell_nof_prb, uint32_t L_prb, uint32_t n_prb0, uint32_t n_prb1, const orc_cf_t* grid,
                           orc_cf_t* ce, orc_chest_ul_res_t* res)
{
  return orc_chest_ul_pusch_hop_cp(r_dmrs, cell_nof_prb, L_prb, n_prb0, n_prb1, 7, grid, ce, res);
}

int orc_chest_ul_pusch_hop_cp(const orc_cf_t* r_dmrs, uint32_t cell_nof_prb, uint32_t L_prb, uint32_t n_prb0, uint32_t n_prb1, uint32_t nsl,
                              const orc_cf_t* grid, orc_cf_t* ce, orc_chest_ul_res_t* res)
{ /* nsl = SRSLTE_CP_NSYMB(cell.cp): the DMRS symbol of slot s is SRSLTE_REFSIGNAL_UL_L(s, cp) = (s + 1) nsl - 4 (refsignal_ul.h:43) */ /* srslte_chest_ul_estimate_pusch (chest_ul.c:268-327) with the defaults of srslte_chest_ul_init (:101-102: 3-tap filter, w = 0.3333),
     no linear interpolation (DO_LINEAR_INTERPOLATION is not defined, :244-258): every slot's estimate is copied over that slot at the
     slot's own PRB offset grant.n_prb[s] - intra-subframe hopping works, upstream only prints a complaint (:293-295) */
  const uint32_t n_prbs[2] = {n_prb0, n_prb1};
  const uint32_t nre = 12 * cell_nof_prb, nrefs = 12 * L_prb;
  cf *           recv = malloc(sizeof(cf) * 2 * nrefs), *est = malloc(sizeof(cf) * 2 * nrefs), *tmp = malloc(sizeof(cf) * nrefs);
  float          filter[3];
  const float    w = 0.3333f;
  filter[0] = w; filter[2] = w; filter[1] = 1 - 2 * w;
  for (uint32_t s = 0; s < 2; s++) {
    const uint32_t L = (s + 1) * nsl - 4; /* SRSLTE_REFSIGNAL_UL_L */
    for (uint32_t i = 0; i < nrefs; i++) {
      recv[s * nrefs + i] = grid[L * nre + n_prbs[s] * 12 + i];
      est[s * nrefs + i]  = c_mulconj(recv[s * nrefs + i], r_dmrs[s * nrefs + i]);
    }
  }
  float power = 0;
  for (uint32_t s = 0; s < 2; s++) {
    const uint32_t L = (s + 1) * nsl - 4;
    cf*            dst = &ce[L * nre + n_prbs[s] * 12];
    conv_same_cf(&est[s * nrefs], filter, dst, nrefs, 3);
    for (uint32_t l = 0; l < nsl; l++) {
      if (s * nsl + l != L) memcpy(&ce[(s * nsl + l) * nre + n_prbs[s] * 12], dst, sizeof(cf) * nrefs);
    }
    for (uint32_t i = 0; i < nrefs; i++) tmp[i] = c_sub(dst[i], est[s * nrefs + i]); /* srslte_chest_estimate_noise_pilots */
    power += avg_power(tmp, nrefs);
  }
  power /= 2;
  float a = (float)(7.419 * w * w + 0.1117 * w - 0.005387); /* chest_ul.c:217-221, "calibrated for filter length 3" */
  res->noise_estimate     = (float)(power / (a * 0.8));
  res->snr                = res->noise_estimate ? avg_power(recv, 2 * nrefs) / res->noise_estimate : NAN;
  res->snr_db             = (float)(10 * log10(res->snr));
  res->noise_estimate_dbm = (float)(10 * log10(res->noise_estimate) + 30);
  res->cfo                = 0;
  free(recv); free(est); free(tmp);
  return 0;
}
