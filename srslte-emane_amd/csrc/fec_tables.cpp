// Host-side 3GPP TS 36.212 tables and index maps used by the FEC kernels of libsrslte_phy_hip.so:
// Table 5.1.3-3 (K, f1, f2), code-block segmentation (5.1.2; replaces cbsegm.c:53-150), the QPP interleaver in
// natural and window-interleaved index space (5.1.3.2.3; replaces tc_interl_lte.c:65-114) and the turbo
// rate-matching circular-buffer order (5.1.4.1; replaces rm_turbo.c:160-233).
#include "phy_hip_internal.hpp"
#include <math.h>
#include <string.h>

const QppRow lte_qpp_table[188] = {
    {40, 3, 10}, {48, 7, 12}, {56, 19, 42}, {64, 7, 16}, {72, 7, 18}, {80, 11, 20},
    {88, 5, 22}, {96, 11, 24}, {104, 7, 26}, {112, 41, 84}, {120, 103, 90}, {128, 15, 32},
    {136, 9, 34}, {144, 17, 108}, {152, 9, 38}, {160, 21, 120}, {168, 101, 84}, {176, 21, 44},
    {184, 57, 46}, {192, 23, 48}, {200, 13, 50}, {208, 27, 52}, {216, 11, 36}, {224, 27, 56},
    {232, 85, 58}, {240, 29, 60}, {248, 33, 62}, {256, 15, 32}, {264, 17, 198}, {272, 33, 68},
    {280, 103, 210}, {288, 19, 36}, {296, 19, 74}, {304, 37, 76}, {312, 19, 78}, {320, 21, 120},
    {328, 21, 82}, {336, 115, 84}, {344, 193, 86}, {352, 21, 44}, {360, 133, 90}, {368, 81, 46},
    {376, 45, 94}, {384, 23, 48}, {392, 243, 98}, {400, 151, 40}, {408, 155, 102}, {416, 25, 52},
    {424, 51, 106}, {432, 47, 72}, {440, 91, 110}, {448, 29, 168}, {456, 29, 114}, {464, 247, 58},
    {472, 29, 118}, {480, 89, 180}, {488, 91, 122}, {496, 157, 62}, {504, 55, 84}, {512, 31, 64},
    {528, 17, 66}, {544, 35, 68}, {560, 227, 420}, {576, 65, 96}, {592, 19, 74}, {608, 37, 76},
    {624, 41, 234}, {640, 39, 80}, {656, 185, 82}, {672, 43, 252}, {688, 21, 86}, {704, 155, 44},
    {720, 79, 120}, {736, 139, 92}, {752, 23, 94}, {768, 217, 48}, {784, 25, 98}, {800, 17, 80},
    {816, 127, 102}, {832, 25, 52}, {848, 239, 106}, {864, 17, 48}, {880, 137, 110}, {896, 215, 112},
    {912, 29, 114}, {928, 15, 58}, {944, 147, 118}, {960, 29, 60}, {976, 59, 122}, {992, 65, 124},
    {1008, 55, 84}, {1024, 31, 64}, {1056, 17, 66}, {1088, 171, 204}, {1120, 67, 140}, {1152, 35, 72},
    {1184, 19, 74}, {1216, 39, 76}, {1248, 19, 78}, {1280, 199, 240}, {1312, 21, 82}, {1344, 211, 252},
    {1376, 21, 86}, {1408, 43, 88}, {1440, 149, 60}, {1472, 45, 92}, {1504, 49, 846}, {1536, 71, 48},
    {1568, 13, 28}, {1600, 17, 80}, {1632, 25, 102}, {1664, 183, 104}, {1696, 55, 954}, {1728, 127, 96},
    {1760, 27, 110}, {1792, 29, 112}, {1824, 29, 114}, {1856, 57, 116}, {1888, 45, 354}, {1920, 31, 120},
    {1952, 59, 610}, {1984, 185, 124}, {2016, 113, 420}, {2048, 31, 64}, {2112, 17, 66}, {2176, 171, 136},
    {2240, 209, 420}, {2304, 253, 216}, {2368, 367, 444}, {2432, 265, 456}, {2496, 181, 468}, {2560, 39, 80},
    {2624, 27, 164}, {2688, 127, 504}, {2752, 143, 172}, {2816, 43, 88}, {2880, 29, 300}, {2944, 45, 92},
    {3008, 157, 188}, {3072, 47, 96}, {3136, 13, 28}, {3200, 111, 240}, {3264, 443, 204}, {3328, 51, 104},
    {3392, 51, 212}, {3456, 451, 192}, {3520, 257, 220}, {3584, 57, 336}, {3648, 313, 228}, {3712, 271, 232},
    {3776, 179, 236}, {3840, 331, 120}, {3904, 363, 244}, {3968, 375, 248}, {4032, 127, 168}, {4096, 31, 64},
    {4160, 33, 130}, {4224, 43, 264}, {4288, 33, 134}, {4352, 477, 408}, {4416, 35, 138}, {4480, 233, 280},
    {4544, 357, 142}, {4608, 337, 480}, {4672, 37, 146}, {4736, 71, 444}, {4800, 71, 120}, {4864, 37, 152},
    {4928, 39, 462}, {4992, 127, 234}, {5056, 39, 158}, {5120, 39, 80}, {5184, 31, 96}, {5248, 113, 902},
    {5312, 41, 166}, {5376, 251, 336}, {5440, 43, 170}, {5504, 21, 86}, {5568, 43, 174}, {5632, 45, 176},
    {5696, 45, 178}, {5760, 161, 120}, {5824, 89, 182}, {5888, 323, 184}, {5952, 47, 186}, {6016, 23, 94},
    {6080, 47, 190}, {6144, 263, 480}};

int lte_cb_index(uint32_t K)
{
  for (int j = 0; j < 188; j++) {
    if (lte_qpp_table[j].K >= K) return j;
  }
  return SRSLTE_ERROR;
}

extern "C" int srslte_hip_cbsegm_cbindex(uint32_t long_cb) { return lte_cb_index(long_cb); } // cbsegm.c:115-126
extern "C" int srslte_hip_cbsegm_cbsize(uint32_t index) { return index < 188 ? (int)lte_qpp_table[index].K : SRSLTE_ERROR; } // cbsegm.c:133-139

extern "C" int srslte_hip_cbsegm(srslte_hip_cbsegm_t* s, uint32_t tbs)
{ // cbsegm.c:53-107
  if (!s) return SRSLTE_ERROR_INVALID_INPUTS;
  memset(s, 0, sizeof(*s));
  if (tbs == 0) return SRSLTE_SUCCESS;
  const uint32_t Z = 6144;
  uint32_t       B = tbs + 24, Bp;
  s->tbs = tbs;
  if (B <= Z) {
    s->C = 1;
    Bp   = B;
  } else {
    s->C = (uint32_t)ceilf((float)B / (Z - 24));
    Bp   = B + 24 * s->C;
  }
  int idx1 = lte_cb_index((Bp - 1) / s->C + 1);
  if (idx1 < 0) return SRSLTE_ERROR;
  s->K1     = lte_qpp_table[idx1].K;
  s->K1_idx = (uint32_t)idx1;
  if (s->C == 1) {
    s->C1 = 1;
  } else {
    if (idx1 == 0) return SRSLTE_ERROR;
    s->K2     = lte_qpp_table[idx1 - 1].K;
    s->K2_idx = (uint32_t)idx1 - 1;
    s->C2     = (s->C * s->K1 - Bp) / (s->K1 - s->K2);
    s->C1     = s->C - s->C2;
  }
  s->F = s->C1 * s->K1 + s->C2 * s->K2 - Bp;
  return SRSLTE_SUCCESS;
}

static inline uint32_t win_of_nat(uint32_t n, uint32_t K, uint32_t W) { return (n % (K / W)) * W + n / (K / W); }
static inline uint32_t nat_of_win(uint32_t x, uint32_t K, uint32_t W) { return (x % W) * (K / W) + x / W; }

void lte_qpp_tables(uint32_t K, uint32_t W, std::vector<uint16_t>& fwd, std::vector<uint16_t>& rev)
{ // pi(i) = (f1 i + f2 i^2) mod K, optionally re-indexed into x = k*W + w  <->  n = w*(K/W) + k
  const int            idx = lte_cb_index(K);
  const uint64_t       f1 = lte_qpp_table[idx].f1, f2 = lte_qpp_table[idx].f2;
  std::vector<uint16_t> f(K), r(K);
  for (uint64_t i = 0; i < K; i++) {
    const uint64_t j = (f1 * i + f2 * i * i) % K;
    f[i]             = (uint16_t)j;
    r[j]             = (uint16_t)i;
  }
  fwd.resize(K);
  rev.resize(K);
  for (uint32_t i = 0; i < K; i++) {
    fwd[i] = W > 1 ? (uint16_t)win_of_nat(f[nat_of_win(i, K, W)], K, W) : f[i];
    rev[i] = W > 1 ? (uint16_t)win_of_nat(r[nat_of_win(i, K, W)], K, W) : r[i];
  }
}

extern "C" int srslte_hip_tc_interl_LTE_gen_interl(uint16_t* forward, uint16_t* reverse, uint32_t long_cb, uint32_t interl_win)
{ // tc_interl_lte.c:75-114
  const int idx = lte_cb_index(long_cb);
  if (!forward || !reverse) return SRSLTE_ERROR_INVALID_INPUTS;
  if (idx < 0 || lte_qpp_table[idx].K != long_cb || (interl_win > 1 && long_cb % interl_win)) {
    hip_log("[srslte_hip] Can't find long_cb=%u in valid TC CB table\n", long_cb);
    return SRSLTE_ERROR;
  }
  std::vector<uint16_t> f, r;
  lte_qpp_tables(long_cb, interl_win, f, r);
  memcpy(forward, f.data(), long_cb * sizeof(uint16_t));
  memcpy(reverse, r.data(), long_cb * sizeof(uint16_t));
  return SRSLTE_SUCCESS;
}

void lte_rm_rx_table(uint32_t K, uint32_t rv, std::vector<uint32_t>& d_index)
{ // 36.212 5.1.4.1.1-2: sub-block interleavers, bit collection, k0; entry n = index 3*i+s of the n-th non-NULL bit read from k0
  static const uint8_t P[32] = {0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30, 1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31};
  const uint32_t D = K + 4, R = (D - 1) / 32 + 1, Kp = 32 * R, ND = Kp - D, Ncb = 3 * Kp;
  std::vector<int32_t> w(Ncb);
  for (uint32_t j = 0; j < 32; j++) {
    for (uint32_t i = 0; i < R; i++) {
      const int32_t  y = (int32_t)(i * 32 + P[j]) - (int32_t)ND;
      const uint32_t k = j * R + i;
      w[k]             = y >= 0 ? 3 * y : -1;
      w[Kp + 2 * k]    = y >= 0 ? 3 * y + 1 : -1;
      const int32_t y2 = (int32_t)((P[k / R] + 32 * (k % R) + 1) % Kp) - (int32_t)ND;
      w[Kp + 2 * k + 1] = y2 >= 0 ? 3 * y2 + 2 : -1;
    }
  }
  const uint32_t k0 = R * (2 * (uint32_t)ceilf((float)Ncb / (float)(8 * R)) * rv + 2);
  d_index.clear();
  for (uint32_t j = 0; d_index.size() < 3 * K + 12; j++) {
    const int32_t d = w[(k0 + j) % Ncb];
    if (d >= 0) d_index.push_back((uint32_t)d);
  }
}

// Gold sequence c(n) of 36.211 7.2 (sequence.c:48-79): x1 from 1 0 0 ..., x2 from c_init, both advanced Nc = 1600 steps
void lte_gold_sequence(uint32_t c_init, uint32_t len, std::vector<uint8_t>& c)
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
