// TEST INFRASTRUCTURE: exercises the pure host side of libsrslte_phy_hip.so (fec_tables.cpp: segmentation, QPP interleaver tables, rate
// de-matching tables) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on
// the pool). Built and run by tests/test_host_sanitizers.py; prints a checksum so that the run cannot be optimised away.
#include "phy_hip_internal.hpp"
#include <stdarg.h>
#include <stdio.h>

void hip_log(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
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
  printf("host sanitizer run ok, checksum %llu\n", sum);
  return 0;
}
