// TEST INFRASTRUCTURE: exercises the pure host side of libsrslte_phy_hip.so (fec_tables.cpp: segmentation, QPP interleaver tables, rate
// de-matching tables) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on
// the pool). Built and run by tests/test_host_sanitizers.py; prints a checksum so that the run cannot be optimised away.
#include "phy_hip_internal.hpp"
#include "srslte_hip/srslte_compat.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

void hip_log(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
}

// compat_refsignal.cpp's two device helpers launch kernels that live in chest.hip; this host-only build never calls them
int chest_average_pilots_launch(const void*, void*, const float*, int, int, int, hipStream_t) { return -1; }
int chest_noise_pilots_launch(const void*, const void*, void*, int, float*, hipStream_t) { return -1; }

// refsignal_dl.h / chest_common.h helpers and the 25.212 interleaver (compat_refsignal.cpp): tables sized exactly as a caller would size
// them, so that any out-of-range index shows up as a heap overflow
static int refsignal_and_filters(unsigned long long* sum)
{
  for (uint32_t prb : {6u, 15u, 25u, 50u, 75u, 100u}) {
    for (uint32_t cp = 0; cp < 2; cp++) {
      srslte_cell_t cell;
      memset(&cell, 0, sizeof(cell));
      cell.nof_prb = prb; cell.nof_ports = 4; cell.id = 17 * prb % 504; cell.cp = cp ? SRSLTE_CP_EXT : SRSLTE_CP_NORM;
      srslte_refsignal_t q;
      if (srslte_refsignal_cs_init(&q, prb) || srslte_refsignal_cs_set_cell(&q, cell)) return 10;
      const uint32_t        nsym = cp ? 12 : 14;
      std::vector<cf_t>     grid((size_t)nsym * 12 * prb), pil(8 * prb);
      memset(grid.data(), 0, sizeof(cf_t) * grid.size());
      srslte_dl_sf_cfg_t sf;
      memset(&sf, 0, sizeof(sf));
      for (uint32_t tti = 0; tti < 10; tti++) {
        sf.tti = tti;
        for (uint32_t port = 0; port < 4; port++) {
          if (srslte_refsignal_cs_put_sf(&q, &sf, port, grid.data()) || srslte_refsignal_cs_get_sf(&q, &sf, port, grid.data(), pil.data())) return 11;
          *sum += srslte_refsignal_cs_nof_re(&q, &sf, port);
        }
      }
      srslte_refsignal_free(&q);
      if (cp) { // MBSFN subframes are extended CP
        srslte_refsignal_t m;
        if (srslte_refsignal_mbsfn_init(&m, prb) || srslte_refsignal_mbsfn_set_cell(&m, cell, (uint16_t)(prb + 3))) return 12;
        std::vector<cf_t> mp(20 * prb), csp(2 * prb);
        memset(csp.data(), 0, sizeof(cf_t) * csp.size());
        if (srslte_refsignal_mbsfn_put_sf(cell, 0, csp.data(), m.pilots[0][3], grid.data()) || srslte_refsignal_mbsfn_get_sf(cell, 0, grid.data(), mp.data())) return 13;
        srslte_refsignal_free(&m);
      }
    }
  }
  for (int n = 1; n <= 16; n++) {
    std::vector<float> f(n);
    if ((int)srslte_chest_set_triangle_filter(f.data(), n) != n) return 14;
    std::vector<float> g(n);
    if ((int)srslte_chest_set_smooth_filter_gauss(g.data(), (uint32_t)n - 1, 1.5f) != n) return 15;
    *sum += (unsigned long long)(1000 * (f[n / 2] + g[n / 2]));
  }
  srslte_tc_interl_t it;
  if (srslte_tc_interl_init(&it, 5114)) return 16;
  for (uint32_t K = 40; K <= 5114; K++) {
    if (srslte_tc_interl_UMTS_gen(&it, K)) return 17;
    *sum += it.forward[K - 1] + it.reverse[K / 2];
  }
  srslte_tc_interl_free(&it);
  srslte_tc_interl_t exact; // tables of exactly K entries for a few sizes, among them the corner cases K = R C and 481..530
  for (uint32_t K : {40u, 159u, 160u, 200u, 201u, 480u, 481u, 530u, 531u, 2280u, 2281u, 2480u, 3160u, 3210u, 5114u}) {
    if (srslte_tc_interl_init(&exact, K) || srslte_tc_interl_UMTS_gen(&exact, K)) return 18;
    srslte_tc_interl_free(&exact);
  }
  return 0;
}

int main(void)
{
  unsigned long long sum = 0;
  for (uint32_t tbs = 16; tbs <= 110000; tbs += (tbs < 7000 ? 8 : 2008)) { // every small size, a sweep of the large ones
    srslte_hip_cbsegm_t s;
    if (srslte_hip_cbsegm(&s, tbs)) return 1;
    sum += s.C * 31 + s.K1 + s.K2 * 7 + s.F;
  }
  for (int idx = 0; idx < 188; idx++) {
    const uint32_t K = lte_qpp_table[idx].K;
    if (lte_cb_index(K) != idx || srslte_hip_cbsegm_cbsize(idx) != (int)K || srslte_hip_cbsegm_cbindex(K) != idx) return 2;
    for (uint32_t W : {0u, 8u, 16u, 32u}) {
      if (W && (K % W)) continue;
      std::vector<uint16_t> f, r;
      lte_qpp_tables(K, W, f, r);
      if (f.size() != K || r.size() != K) return 3;
      for (uint32_t i = 0; i < K; i++) {
        if (f[i] >= K || r[i] >= K) return 4;
      }
      sum += f[K / 2] + r[K / 3];
    }
    for (uint32_t rv = 0; rv < 4; rv++) {
      std::vector<uint32_t> t;
      lte_rm_rx_table(K, rv, t);
      for (uint32_t v : t) {
        if (v >= 3 * K + 12) return 5;
      }
      sum += t.size() + t[t.size() / 2];
    }
    std::vector<uint16_t> fw(K + 16), rv_(K + 16);
    if (srslte_hip_tc_interl_LTE_gen_interl(fw.data(), rv_.data(), K, 0)) return 6;
    sum += fw[1];
  }
  if (int rc = refsignal_and_filters(&sum)) return rc;
  printf("host sanitizer run ok, checksum %llu\n", sum);
  return 0;
}
