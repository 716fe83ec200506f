// LTE turbo decoder for gfx950: srslte_tdec_run_all / srslte_tdec_iteration (turbodecoder.c:539-562), bit-exact with the
// reference's 16-bit back-ends as selected on an AVX2 host (turbodecoder.c:394-420):
//   K > 800, K%16==0 : 16-window max-log-MAP, saturating int16, normalise every 2 steps          (turbodecoder_win.h, avx16)
//   K > 400, K%8==0  :  8-window, same but extrinsic output >> 1                                  (turbodecoder_win.h, sse16)
//   otherwise        : unwindowed, wrapping int16, normalise every 4 steps, tail inside the sweep (turbodecoder_gen.c)
// and the extrinsic-exchange schedule of turbodecoder_iter.h:71-139 with CRC early stop as sch.c:353-383.
//
// Mapping to CDNA4 (no MFMA: there is no contraction here):
//   * one 64-lane wavefront per code block (windowed) / per 8 code blocks (generic);
//   * a lane owns ONE trellis state of TWO windows: the two int16 path metrics are packed in one VGPR and every
//     add/sub/max is a single v_pk_{add,sub}_i16 [clamp] / v_pk_max_i16, i.e. exactly _mm256_adds_epi16 & co;
//   * lanes are (state-slot p: lane bits 0,1,3) x (window pair g: lane bits 2,4,5). The radix-2 trellis butterfly
//     {2m,2m+1} -> {m,m+4} is done IN PLACE: after a step the slot that held state s holds state rotr3(s), so at
//     time t slot p holds state rotr3^t(p) and its butterfly partner is always p ^ (1 << (t%3)), i.e. lane ^1,
//     lane ^2 or lane ^8 - one DPP move (quad_perm / row_ror:8), no LDS, no ds_bpermute in the inner loop;
//   * the backward (beta) metrics of all steps are parked in a per-wave global scratch slab, one coalesced
//     256-B store/load per step (HBM/L2; 8*K*2 B per block is more LDS than a CU can spare at useful occupancy);
//   * 40-step warm-ups from -INF over the neighbouring window, the one-window shift of the start metrics and the
//     scalar tail trellis are as turbodecoder_win.h:398-456,:529-583,:351-395.
#include "common.hpp"
#include "phy_hip_internal.hpp"
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

namespace {

typedef short v2s __attribute__((ext_vector_type(2)));
typedef int   pk_t; // two int16 lanes: lo = first window of the pair, hi = second

#ifndef TDEC_EWU
#define TDEC_EWU 4 // 16-byte elements in flight per lane in the element-wise phases
#endif
// -DTDEC_PROF: per-phase s_memtime accounting of the windowed kernels (diagnostic build only; stamps cost ~10 %)
#ifdef TDEC_PROF
#define PROF_DECL unsigned long long prof_t0 = __builtin_readcyclecounter(), prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(i)                                               \
  {                                                           \
    const unsigned long long t_ = __builtin_readcyclecounter(); \
    prof_acc[i] += t_ - prof_t0;                              \
    prof_t0 = t_;                                             \
  }
#define PROF_ARGS , unsigned long long& prof_t0, unsigned long long* prof_acc
#define PROF_PASS , prof_t0, prof_acc
#else
#define PROF_DECL
#define PROF(i)
#define PROF_ARGS
#define PROF_PASS
#endif
constexpr int TD_INF      = 10000;
constexpr int SRSLTE_HIP_MAX_K = 6144;
constexpr int WIN_OVERLAP = 40;
constexpr int CKPT        = 24; // beta checkpoint spacing = alpha segment length (multiple of 6)

__device__ __forceinline__ v2s  as_v(pk_t a) { return __builtin_bit_cast(v2s, a); }
__device__ __forceinline__ pk_t as_p(v2s a) { return __builtin_bit_cast(pk_t, a); }
template <bool SAT>
__device__ __forceinline__ pk_t pk_add(pk_t a, pk_t b)
{
  if constexpr (SAT) return as_p(__builtin_elementwise_add_sat(as_v(a), as_v(b)));
  return as_p(as_v(a) + as_v(b));
}
template <bool SAT>
__device__ __forceinline__ pk_t pk_sub(pk_t a, pk_t b)
{
  if constexpr (SAT) return as_p(__builtin_elementwise_sub_sat(as_v(a), as_v(b)));
  return as_p(as_v(a) - as_v(b));
}
__device__ __forceinline__ pk_t pk_max(pk_t a, pk_t b) { return as_p(__builtin_elementwise_max(as_v(a), as_v(b))); }
// AR = 1: the 8-bit back-ends (sse8/avx8). An int8 metric lives in the HIGH byte of its int16 half (value << 8): the packed
// int16 saturating add/sub then saturates exactly where _mm_adds_epi8/_mm_subs_epi8 do, except that the positive limit
// comes out as 0x7fff instead of 0x7f00 - one v_and restores it. max needs nothing. A stray 0xff low byte is harmless as long
// as the other operand of the next add has a clean low byte (no carry into the metric) and the value is masked before it is
// subtracted or stored: the trellis step therefore masks only the new state metric and the two reduced LLR candidates.
constexpr pk_t M8 = (pk_t)0xFF00FF00;
template <int AR>
__device__ __forceinline__ pk_t s_add(pk_t a, pk_t b)
{
  const pk_t r = pk_add<true>(a, b);
  return AR ? (r & M8) : r;
}
template <int AR>
__device__ __forceinline__ pk_t s_sub(pk_t a, pk_t b)
{
  const pk_t r = pk_sub<true>(a, b);
  return AR ? (r & M8) : r;
}
__device__ __forceinline__ pk_t pk_make(int lo, int hi) { return (pk_t)((lo & 0xffff) | (hi << 16)); }
__device__ __forceinline__ int  pk_lo(pk_t a) { return (int)(short)(a & 0xffff); }
__device__ __forceinline__ int  pk_hi(pk_t a) { return a >> 16; }

// A DPP move whose every destination lane has a source lane (quad_perm, row_ror): no `old` operand, so the compiler does not
// have to copy the source into the destination first.
#define DPP_MOV(v, ctrl) __builtin_amdgcn_mov_dpp((v), (ctrl), 0xf, 0xf, false)
// lane ^ 1, ^ 2, ^ 8 as DPP moves
template <int PH>
__device__ __forceinline__ pk_t dpp_partner(pk_t v)
{
  if constexpr (PH == 0) return DPP_MOV(v, 0xB1);  // quad_perm [1,0,3,2]
  if constexpr (PH == 1) return DPP_MOV(v, 0x4E);  // quad_perm [2,3,0,1]
  return DPP_MOV(v, 0x128);                        // row_ror:8
}
// value of slot 0 (state 0 at every time step) broadcast to the 8 slots of the group
__device__ __forceinline__ pk_t bcast_slot0(pk_t v)
{
  pk_t t = DPP_MOV(v, 0x00);                                          // quad_perm [0,0,0,0]
  return __builtin_amdgcn_update_dpp(t, t, 0x118, 0xf, 0xf, false);   // row_shr:8, lanes 0..7 of the row keep t
}

struct LaneGeom {
  int  lane, p, g;
  bool p0, p1, p2;
  int  m0, m1, m2; // all-ones where slot bit 0 / 1 / 2 is set: VGPR masks for v_bfi selections (no SGPR mask traffic)
  int  io[3];      // per trellis phase: which of the four branch metrics {0, x, y, x+y} is this slot's own one
  int  mk0, mk1;   // LLR reduction (win_step): b0 ^ lane-bit-0 at phase 0 / 1 (phase 2: always 0)
};
__device__ __forceinline__ LaneGeom lane_geom()
{
  LaneGeom L;
  L.lane = threadIdx.x & 63;
  L.p0   = L.lane & 1;
  L.p1   = (L.lane >> 1) & 1;
  L.p2   = (L.lane >> 3) & 1;
  L.p    = (L.lane & 3) | (((L.lane >> 3) & 1) << 2);
  L.g    = ((L.lane >> 2) & 1) | (((L.lane >> 4) & 3) << 1);
  L.m0   = L.p0 ? -1 : 0;
  L.m1   = L.p1 ? -1 : 0;
  L.m2   = L.p2 ? -1 : 0;
  // (b1, b0) = state bits 2, 1 of the slot's state at the phase (see acs): own metric index 2*b1 + b0
  L.io[0] = 2 * L.p2 + L.p1;
  L.io[1] = 2 * L.p0 + L.p2;
  L.io[2] = 2 * L.p1 + L.p0;
  L.mk0   = L.m1 ^ L.m0;
  L.mk1   = L.m2 ^ L.m0;
  return L;
}
__device__ __forceinline__ int lane_of(int g, int p) { return (p & 3) | ((g & 1) << 2) | (((p >> 2) & 1) << 3) | ((g >> 1) << 4); }
__device__ __forceinline__ int rotr3(int s, int n)
{
  for (int i = 0; i < n; i++) s = (s >> 1) | ((s & 1) << 2);
  return s;
}
__device__ __forceinline__ int rotl3(int s, int n)
{
  for (int i = 0; i < n; i++) s = ((s << 1) & 7) | (s >> 2);
  return s;
}

// One add-compare-select step for the slot's state at phase PH = time % 3 (time of the OLD metrics for the forward
// recursion, of the NEW metrics for the backward one - the same code serves both, see header comment).
// Branch metrics: (own, partner) = (0,xy) (x,y) (y,x) (xy,0) for state>>1 = 0..3 (turbodecoder_win.h:471-491,:621-641).
// Windowed decoders: the combine pass stores {0, x, y, x+y} per step and each lane loads its own branch metric g_own
// (index LaneGeom::io[PH]); the partner's is the complementary entry, held by the lane whose two (b1, b0) slot bits are
// flipped: one or two DPP moves instead of six select operations.
template <int PH>
__device__ __forceinline__ pk_t dpp_complement(pk_t g)
{
  if constexpr (PH == 2) return DPP_MOV(g, 0x1B);          // slot bits 1,0: quad_perm [3,2,1,0]
  const pk_t t = PH == 0 ? DPP_MOV(g, 0x4E) : DPP_MOV(g, 0xB1); // slot bit 1 (lane ^ 2) / slot bit 0 (lane ^ 1)
  return DPP_MOV(t, 0x128);                                // slot bit 2 (lane ^ 8)
}
template <int PH, int AR>
__device__ __forceinline__ pk_t acs_tab(pk_t old, pk_t g_own, pk_t* t_own, pk_t* t_par)
{
  const pk_t g_par = dpp_complement<PH>(g_own);
  const pk_t po    = dpp_partner<PH>(old);
  *t_own           = pk_add<true>(old, g_own);
  *t_par           = pk_add<true>(po, g_par);
  const pk_t nv    = pk_max(*t_own, *t_par);
  return AR ? (nv & M8) : nv;
}

template <int PH, bool SAT, int AR = 0>
__device__ __forceinline__ pk_t acs(const LaneGeom& L, pk_t old, pk_t x, pk_t y, pk_t xy, pk_t* t_own, pk_t* t_par)
{
  const int  b1 = PH == 0 ? L.m2 : (PH == 1 ? L.m0 : L.m1); // state bit 2 of this slot at this phase, as a mask
  const int  b0 = PH == 0 ? L.m1 : (PH == 1 ? L.m2 : L.m0); // state bit 1
#define BFI(m, a, b) (((a) & (m)) | ((b) & ~(m)))               /* v_bfi_b32 */
  const pk_t g_own = BFI(b1, BFI(b0, xy, y), x & b0);
  const pk_t g_par = BFI(b1, x & ~b0, BFI(b0, y, xy));
#undef BFI
  const pk_t po    = dpp_partner<PH>(old);
  *t_own = pk_add<SAT>(old, g_own);
  *t_par = pk_add<SAT>(po, g_par);
  const pk_t nv = pk_max(*t_own, *t_par);
  return AR ? (nv & M8) : nv;
}

// max over the 8 slots of a group
__device__ __forceinline__ pk_t group_max(pk_t v)
{
  v = pk_max(v, dpp_partner<0>(v));
  v = pk_max(v, dpp_partner<1>(v));
  v = pk_max(v, dpp_partner<2>(v));
  return v;
}

struct TdecTables {            // per (K, W) device tables
  const uint16_t* inter;       // app1[inter[i]] = ext2[i]
  const uint16_t* deinter;     // app2[deinter[i]] = ext1[i]
  const uint32_t* crc_rem;     // x^(nbits-1-nat(i)) mod g for array position i (0 beyond nbits); nullptr = no early stop
  const uint32_t* crc_rem_i;   // crc_rem[inter[j]]: the same remainders for position j of the INTERLEAVED sequence (tdec_pair_kernel)
};

struct TdecArgs {
  const int16_t* in;
  uint32_t       in_stride;
  int            sb_layout;
  uint32_t       K, nof_cb, nof_iter;
  int16_t*       work;         // per block: 7 arrays of Kp int16
  uint32_t       Kp;
  pk_t*          beta;         // per wave: (steps+1) * 64 dwords
  int4*          xy;           // per block: K x 16 B (combine-pass scratch of the windowed decoders: x, y, x + y arrays)
  const pk_t*    zeros;        // max_long_cb zero dwords (the "0" branch metric, read like the other three)
  uint32_t       beta_stride;  // dwords per wave
  uint8_t*       out;
  uint32_t       out_stride;
  uint32_t*      iters;
  uint8_t*       crc_ok;
  TdecTables     t;
  const uint32_t* tb_rem;      // optional [tb_C][K] remainders for the TB CRC share of each block (array order), else nullptr
  uint32_t        tb_C;
  uint32_t*       tb_syn;      // [nof_cb] out
  uint32_t        start_iter;  // passes 0..start_iter-1 were run by the previous launch on these blocks (same input, work arrays untouched): resume
  const uint8_t*  skip;        // optional [nof_cb]: blocks whose CRC passed in an earlier transmission keep their bytes and flags (sch.c:317-318)
  const uint32_t* cb_map;      // optional [nof_cb]: launched block i works on block slot cb_map[i] of in / out / iters / crc_ok / skip (ragged
                               // batches: the blocks of one length are scattered over the batch); the work arrays stay per launched block
  // tdec_pair_kernel, optional: the blocks of transport block cb / tb_C write their payload bytes straight into it (sch.c:360,:401-410) and
  // the last one to finish gives the verdict of sch.c:470-488 - no assembly kernel behind the decoder (tdec_set_tb_direct)
  uint8_t*        tb_out;      // [nof_cb / tb_C][tb_out_stride] or nullptr
  uint32_t        tb_out_stride, tb_rb; // tb_rb: payload bytes per block (K / 8 - 3 with several blocks, K / 8 with one)
  uint8_t*        tb_ok_out;   // [nof_cb / tb_C]
  uint32_t*       tb_acc;      // [nof_cb / tb_C][4]: syndrome xor, flags, finished blocks, -; zero between launches (the last block clears them)
  unsigned long long* prof;    // -DTDEC_PROF builds: [nof_cb][10] cycles per phase, else unused
  int            dbg;          // timing experiments only (SRSLTE_HIP_TDEC_DBG): 1 skips the SISO sweeps, 2 the element-wise subtractions,
                               // 4 replaces the interleaver scatters by in-order stores, 8 points every branch-metric load at the zero buffer
  // tdec_mix_kernel only (behind everything the other kernels read), with tb_out: transport blocks of DIFFERENT sizes (a ragged batch, tdec_set_tb_ragged).
  // Block slot cb belongs to transport-block slot cb / tb_C (tb_C = the slots' width), which has tb_Cof[cb / tb_C] blocks; its row of tb_out /
  // tb_ok_out is the slot itself below tb_B, tb_rows0 + slot - tb_B from there on (second codewords). tbA: the group's CRC24A tables (tdec_tbA_table)
  const uint8_t*  tb_Cof;
  uint32_t        tb_B, tb_rows0;
  const uint32_t* tbA;
};

// Several block lengths in ONE launch (ragged batches: tdec_run_groups). A launch per length is a serial chain of latency-bound kernels - a few
// blocks each, up to six passes - and that chain, not the work, is what a mixed batch then takes. The table travels by value as a kernel
// argument of its own (read with scalar loads at a dynamic index; TdecArgs stays free of arrays, so its patched copy lives in registers).
constexpr int TDEC_MAX_GROUPS = 24;
struct TdecGroup {
  uint32_t   K, nof_cb, first_lcb, first_wave; // block length; blocks; first launched block / first wavefront of the group within the launch
  TdecTables t;
};
struct TdecGroups {
  uint32_t  n; // 0: a launch of one length, described by TdecArgs alone
  uint32_t  xy_stride;
  TdecGroup g[TDEC_MAX_GROUPS];
  uint8_t   kind[TDEC_MAX_GROUPS]; // tdec_mix_kernel only (behind everything the other kernels read): which decoder the group's blocks take
  const uint32_t* tbA[TDEC_MAX_GROUPS]; // ... and, with tb_Cof, the group's CRC24A share tables: [K] whole block, [K] payload only, [16] x^(j (K - 24))
};
enum { TDEC_KIND_PAIR = 0, TDEC_KIND_WIN8 = 1, TDEC_KIND_GEN = 2 };
// Returns this wavefront's index within its group and makes `a` describe that group alone: its length and tables, its share of the block map,
// of the work arrays (7 Kp per launched block), of the combine rows and of the checkpoint / beta rows (beta_stride per wavefront).
__device__ __forceinline__ uint32_t tdec_enter_group(TdecArgs& a, const TdecGroups& gs)
{
  uint32_t bx = blockIdx.x;
  if (gs.n) {
    uint32_t gi = 0;
    for (uint32_t i = 1; i < gs.n; i++) gi = bx >= gs.g[i].first_wave ? i : gi; // ascending first_wave
    const uint32_t first_lcb = gs.g[gi].first_lcb, first_wave = gs.g[gi].first_wave;
    bx -= first_wave;
    a.K      = gs.g[gi].K;
    a.nof_cb = gs.g[gi].nof_cb;
    a.t      = gs.g[gi].t;
    if (a.cb_map) a.cb_map += first_lcb;
    a.work += (size_t)first_lcb * 7 * a.Kp;
    a.xy += (size_t)first_lcb * gs.xy_stride;
    a.beta += (size_t)first_wave * a.beta_stride;
  }
  return bx;
}

__device__ __forceinline__ uint32_t tdec_gf24_mul(uint32_t a, uint32_t b)
{ // a(x) b(x) mod g_CRC24A
  uint32_t r = 0;
  for (int i = 23; i >= 0; i--) {
    r <<= 1;
    if (r & 0x1000000u) r ^= 0x1864CFBu;
    if ((b >> i) & 1) r ^= a;
  }
  return r;
}

#ifdef TDEC_MIX_TU
// Ragged batches (tdec_set_tb_ragged) with HARQ: a block whose CRC passed in an earlier transmission is not decoded again (sch.c:317-318) - its bytes
// are in its row of a.out since then. It still owes its transport block what a decoded block gives in its last phase: its bytes in the block's
// place, the CRC24A share of those bytes (table lookups in natural bit order: array position (n % Lw) * 16 + n / Lw) and its arrival. What it
// does not give is the "a block was decoded in this call" flag (bit 2): a transport block whose blocks had all passed before is not delivered
// again (the duplicate retransmission of sch.c:404-410 / tb_any_block_decoded in the assembly kernel).
__device__ __forceinline__ void tdec_tb_stored_block(const TdecArgs& a, int cb, int K, int lane)
{
  const int      K8 = K / 8, Lw = K / 16;
  const int      tbr = (int)(cb % a.tb_C), tbi = (int)(cb / a.tb_C), tbC = (int)a.tb_Cof[tbi];
  const int      rb = (tbC == 1 ? K : K - 24) / 8, tlim = tbr == tbC - 1 ? K8 : rb;
  const size_t   row = (uint32_t)tbi < a.tb_B ? (size_t)tbi : (size_t)a.tb_rows0 + tbi - a.tb_B;
  const uint8_t* o   = a.out + (size_t)cb * a.out_stride;
  uint8_t*       tbo = a.tb_out + row * a.tb_out_stride + (size_t)tbr * rb;
  const uint32_t* tab = a.tbA + (tbC == 1 ? 0 : K);
  const uint32_t magic = (uint32_t)((0x100000000ull + (uint32_t)Lw - 1) / (uint32_t)Lw);
  uint32_t       tsyn = 0;
  bool           par_nz = false;
  for (int bi = lane; bi < K8; bi += 64) {
    const uint32_t byte = o[bi];
    if (bi < tlim) tbo[bi] = (uint8_t)byte;
    par_nz = par_nz || (bi >= rb - 3 && bi < rb && byte != 0);
    int q = (int)__umulhi((uint32_t)(8 * bi), magic), r = 8 * bi - q * Lw;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (byte & (0x80u >> j)) tsyn ^= tab[r * 16 + q];
      if (++r == Lw) {
        r = 0;
        q++;
      }
    }
  }
  for (int o2 = 32; o2 > 0; o2 >>= 1) tsyn ^= __shfl_xor(tsyn, o2, 64);
  par_nz = __ballot(par_nz) != 0;
  if (lane == 0) {
    const uint32_t fac = a.tbA[2 * K + (tbC - 1 - tbr)];
    if (fac != 1) tsyn = tdec_gf24_mul(tsyn, fac);
    uint32_t*      acc = a.tb_acc + 4 * (size_t)tbi;
    const uint32_t o1 = atomicXor(acc, tsyn);
    const uint32_t o2 = atomicOr(acc + 1, (tbr == tbC - 1 && par_nz) ? 2u : 0u);
    uint32_t       dep = o1 | o2;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(dep)::"memory");
    if (atomicAdd(acc + 2, 1u) == (uint32_t)tbC - 1) {
      const uint32_t syn = atomicExch(acc, 0u), fl = atomicExch(acc + 1, 0u);
      atomicExch(acc + 2, 0u);
      a.tb_ok_out[row] = (syn == 0 && fl == 6u) ? 1 : 0;
    }
  }
}
#endif

// ------------------------------------------------------------------------------------------------------------------
// Windowed SISO (W = 16: packed pairs w = 2g+h; W = 8: w = g in the low half, high half idle)
// ------------------------------------------------------------------------------------------------------------------
// i-th window pair of a window-interleaved int16 array (i = step * pairs-per-step + pair). AR: the array holds int8 values in
// int16 containers; they move to the high bytes.
template <int W, int AR>
__device__ __forceinline__ pk_t ld_pair(const int16_t* a, int i)
{
  if constexpr (W == 8) return (pk_t)(uint16_t)a[i];
  const pk_t v = reinterpret_cast<const pk_t*>(a)[i];
  return AR ? ((v & 0x00FF00FF) << 8) : v;
}

// ---- one trellis step on prepared inputs: in.x = systematic (+ a-priori, already added with saturation), in.y = parity
// MODE 0: warm-up, 1: beta main pass, 2: alpha main pass (also forms the extrinsic output o from beta value B)
// JPAR: parity of the slot that keeps this step's extrinsic output (MODE 2): only lanes with that lane-bit-0 get a valid o.
template <int PH, int MODE, int W, int AR, int JPAR = 0>
__device__ __forceinline__ void win_step(const LaneGeom& L, pk_t& v, pk_t g_own, pk_t B, pk_t& o)
{
  pk_t to, tp;
  v = acs_tab<PH, AR>(v, g_own, &to, &tp);
  if constexpr (MODE == 2) {
    // Max-log LLR (turbodecoder_win.h:643-659): m0/m1 = max over the 8 states of beta + the transition with info bit 0/1.
    // The own transition carries info bit b0. Lanes with lane bit 0 clear reduce the bit-0 candidates, the others the bit-1
    // ones: each lane keeps one candidate and hands the other to its lane^1 neighbour, so one reduction serves both.
    pk_t keep = to, send = tp; // phase 2: b0 is lane bit 0 itself
    if constexpr (PH != 2) {
      const int q = PH == 0 ? L.mk0 : L.mk1; // b0 ^ lane bit 0
      keep        = (tp & q) | (to & ~q);
      send        = (to & q) | (tp & ~q);
    }
    pk_t r = pk_max(pk_add<true>(B, keep), dpp_partner<0>(pk_add<true>(B, send)));
    r      = pk_max(r, dpp_partner<1>(r));
    r      = pk_max(r, dpp_partner<2>(r));
    if constexpr (AR) r &= M8;
    const pk_t other = dpp_partner<0>(r);
    o                = JPAR ? pk_sub<true>(r, other) : pk_sub<true>(other, r); // m1 - m0
    if constexpr (W == 8) o = as_p(as_v(o) >> (short)1); // divide_output, turbodecoder_win.h:56,:657-659
    // AR: back to an int8 value in an int16 container and divide_output (>> 1, :143-147) in one shift; the 0x7fff of a
    // positive saturation gives 63 = 127 >> 1 without a mask
    if constexpr (AR) o = as_p(as_v(o) >> (short)9);
  }
}

// turbodecoder_win.h:332-349: subtract state 0 (16-bit) / the maximum over the 8 states (normalize_max, 8-bit)
template <int AR>
__device__ __forceinline__ void win_normalize(pk_t& v)
{
  if constexpr (AR) {
    v = pk_sub<true>(v, group_max(v)); // v <= max: the difference never saturates upwards, no mask
  } else {
    v = pk_sub<true>(v, bcast_slot0(v));
  }
}

// gg = window pair within the step (0..7, or 0..15 for 32 windows)
template <int W>
__device__ __forceinline__ void store_out(int16_t* out, int k, int gg, pk_t o)
{
  if constexpr (W == 8) {
    out[k * 8 + gg] = (int16_t)pk_lo(o);
  } else {
    reinterpret_cast<pk_t*>(out)[k * (W / 2) + gg] = o;
  }
}

// Branch metrics through LDS. The four metrics of a step - 0, x, y, x + y - are arrays in the decoder's native order
// (step * G + pair); a lane needs ONE of them per step, which one depends on its slot and the phase. A CU issues one wavefront
// memory instruction per 4 cycles for all four SIMDs, so a load per lane and step makes the sweeps compete for that port with
// every other wavefront. Instead the wave copies the rows of up to 24 steps (8 pairs x 3 arrays, contiguous or nearly so) into
// LDS with three 16-byte loads per lane, one block ahead of their use, and every lane then picks its metric with a ds_read
// at a compile-time offset from a per-lane, per-phase base (the zero metric is a constant LDS region).
constexpr int ST_STEPS = 24;
constexpr int ST_ARR   = ST_STEPS * 8;            // dwords of one array in one buffer: 24 steps x 8 window pairs
constexpr int ST_BUF   = 3 * ST_ARR;              // x, y, x + y
constexpr int ST_TOTAL = 2 * ST_BUF + ST_ARR;     // two buffers + zeros
template <int W>
constexpr int pairs_per_step() { return W == 32 ? 16 : 8; }
typedef int v8i __attribute__((ext_vector_type(8)));
struct Src {                 // global arrays of one SISO pass, offset to the first pair of the half being processed
  const pk_t *x, *y;         // x + y is not an array: it is formed when the rows are staged (one saturating add per dword)
};
struct StRegs {
  int4 a, b;
};
struct Stage {               // per-lane view of the staging area
  pk_t* lds;
  v8i   rb;                  // [buffer * 3 + trellis phase] read base (dword index): array of the slot's own metric + own pair. A vector
                             // VALUE, not an int array: indexing an array with the run-time buffer number turns the struct into an
                             // alloca that the compiler parks in LDS (2.5 KB per wavefront and a ds_read per use) or in scratch
  int   sj, sp;              // copy role: row (step) lane >> 1, half row lane & 1
};
__device__ __forceinline__ void stage_init(Stage& st, pk_t* lds, const LaneGeom& L)
{
  st.lds = lds;
  st.sj  = L.lane >> 1;
  st.sp  = L.lane & 1;
  for (int i = L.lane; i < ST_ARR; i += 64) lds[2 * ST_BUF + i] = 0;
#pragma unroll
  for (int b = 0; b < 2; b++) {
#pragma unroll
    for (int ph = 0; ph < 3; ph++) {
      const int ix = L.io[ph];
      st.rb[b * 3 + ph] = (ix == 0 ? 2 * ST_BUF : b * ST_BUF + (ix - 1) * ST_ARR) + L.g;
    }
  }
}
// rows k_lo .. k_lo + n - 1 (n <= 24) of the three arrays; lanes beyond 2n re-read the last row (no branch around the loads)
template <int G>
__device__ __forceinline__ StRegs stage_load(const Src& S, const Stage& st, int k_lo, int n)
{
  const int off = (k_lo + min(st.sj, n - 1)) * G + st.sp * 4;
  StRegs    r;
  r.a = *reinterpret_cast<const int4*>(S.x + off);
  r.b = *reinterpret_cast<const int4*>(S.y + off);
  return r;
}
// AR (the 8-bit back-ends with 16 / 32 windows): the y rows come straight from the parity array (int8 values in int16 containers) and move
// to the high bytes HERE, where the row is consumed - not behind the load, which would wait for it (the rows are requested two blocks
// ahead). The combine pass then reads and writes one array less. (Forming x = sat(a-priori + systematic) here as well, i.e. no combine
// pass at all, is bit-exact too but takes the kernel from 168 to 242 VGPRs: profiles/r03/ab_llr8_staging.txt.)
template <int AR = 0>
__device__ __forceinline__ void stage_store(const Stage& st, int buf, int n, const StRegs& r)
{
  if (st.sj < n) {
    pk_t*      d = st.lds + buf * ST_BUF + st.sj * 8 + st.sp * 4;
    const int4 y = AR ? make_int4((r.b.x & 0x00FF00FF) << 8, (r.b.y & 0x00FF00FF) << 8, (r.b.z & 0x00FF00FF) << 8, (r.b.w & 0x00FF00FF) << 8) : r.b;
    *reinterpret_cast<int4*>(d)              = r.a;
    *reinterpret_cast<int4*>(d + ST_ARR)     = y;
    // x + y (turbodecoder_win.h:472-491), saturating; for the 8-bit back-ends a 0x7fff here is harmless (see M8)
    *reinterpret_cast<int4*>(d + 2 * ST_ARR) = make_int4(pk_add<true>(r.a.x, y.x), pk_add<true>(r.a.y, y.y), pk_add<true>(r.a.z, y.z), pk_add<true>(r.a.w, y.w));
  }
  // One wavefront per workgroup and the LDS executes a wave's instructions in order: the reads that follow see these writes.
  // Only the compiler has to be kept from reordering them; a __syncthreads() here would also drain vmcnt, i.e. wait for the
  // row loads that were just put in flight for the next blocks.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// metric of the step staged at row jj, phase ph
__device__ __forceinline__ pk_t stage_get(const Stage& st, int buf, int ph, int jj)
{
  const v8i r = st.rb; // constant-index extracts and selects only: a run-time vector index goes through scratch
  const int b0 = buf ? r[3] : r[0], b1 = buf ? r[4] : r[1], b2 = buf ? r[5] : r[2];
  return st.lds[(ph == 0 ? b0 : (ph == 1 ? b1 : b2)) + jj * 8];
}

// nb blocks of BLK steps (BLK a multiple of 6 so that trellis phase and normalisation parity are compile-time), first
// step k_first, direction DIR, phase of the first step PH0, normalisation counter n_first (parity NPAR0).
// Every operand of a block is requested at its top and consumed in order: the memory round trip is paid once per BLK
// steps instead of once per step, and the second wavefront of the SIMD computes meanwhile. (Prefetching across the
// loop back-edge does not survive hipcc's waitcnt insertion, which drains vmcnt(0) there; in-flight registers managed
// by hand from inline asm were tried and are unsafe at this register pressure: the allocator copies them.)
template <int W, int BLK, int PH0, int DIR, int MODE, int NPAR0, int AR>
__device__ __forceinline__ void win_run(const LaneGeom& L, pk_t& v, const Src& S, const Stage& st, int k_first, int n_first, int nb,
                                        pk_t* __restrict__ beta)
{
  static_assert(MODE != 2, "the alpha main pass has its own loop (win_siso)");
  constexpr int G = pairs_per_step<W>();
  if (nb <= 0) return;
  // staged rows of block b: [k0 - (BLK-1), k0] (DIR < 0) or [k0, k0 + BLK) (DIR > 0); row jj of step j
  auto   lo  = [&](int b) { return DIR > 0 ? k_first + BLK * b : k_first - BLK * b - (BLK - 1); };
  // two blocks of rows in flight: a block computes in ~1500 cycles at two waves per SIMD, a load from HBM takes longer
  StRegs r = stage_load<G>(S, st, lo(0), BLK), r2 = r;
  if (nb > 1) r2 = stage_load<G>(S, st, lo(1), BLK);
  for (int b = 0; b < nb; b++) {
    const int k0 = k_first + DIR * BLK * b, n0 = n_first + DIR * BLK * b, buf = b & 1;
    stage_store<(W == 8 ? 0 : AR)>(st, buf, BLK, r);
    r = r2;
    if (b + 2 < nb) r2 = stage_load<G>(S, st, lo(b + 2), BLK);
    pk_t c[BLK];
#pragma unroll
    for (int j = 0; j < BLK; j++) {
      const int ph = DIR > 0 ? (PH0 + j) % 3 : (PH0 + 3 * BLK - j) % 3; // static after unrolling
      c[j]         = st.lds[(buf ? st.rb[3 + ph] : st.rb[ph]) + (DIR > 0 ? j : BLK - 1 - j) * 8];
    }
#pragma unroll
    for (int j6 = 0; j6 < BLK; j6 += 6) {
#define STEP6(J)                                                                  \
  {                                                                               \
    constexpr int PH = DIR > 0 ? (PH0 + J) % 3 : (PH0 + 18 - J) % 3;              \
    pk_t          o  = 0;                                                         \
    win_step<PH, MODE, W, AR, (J & 1)>(L, v, c[j6 + J], 0, o);                    \
    if constexpr (MODE == 1) {                                                    \
      const int kq = k0 + DIR * (j6 + J);                                         \
      if constexpr (BLK == CKPT) { /* aligned blocks end on a multiple of CKPT */ \
        if (j6 + J == BLK - 1) beta[(kq / CKPT) * 64 + L.lane] = v;               \
      } else {                                                                    \
        if (kq % CKPT == 0) beta[(kq / CKPT) * 64 + L.lane] = v;                  \
      }                                                                           \
    }                                                                             \
    if constexpr (AR || ((NPAR0 + J) & 1) == 0) { /* 8-bit: every step; 16-bit: every second one */ \
      /* no normalisation at counter 0; in the beta main pass that is the very last step, whose result nobody reads */ \
      if (MODE == 1 || n0 + DIR * (j6 + J) != 0) win_normalize<AR>(v);            \
    }                                                                             \
  }
      STEP6(0) STEP6(1) STEP6(2) STEP6(3) STEP6(4) STEP6(5)
#undef STEP6
    }
  }
}

// up to 5 left-over steps with run-time phase; their loads are issued together up front
template <int W, int DIR, int MODE, int AR>
__device__ __forceinline__ void win_rem(const LaneGeom& L, pk_t& v, const Src& S, const Stage& st, int k_first, int n_first, int ph_first,
                                        int r, pk_t* __restrict__ beta)
{
  static_assert(MODE != 2, "the alpha main pass has its own loop (win_siso)");
  constexpr int G = pairs_per_step<W>();
  if (r <= 0) return;
  const int k_lo = DIR > 0 ? k_first : k_first - (r - 1);
  stage_store<(W == 8 ? 0 : AR)>(st, 0, r, stage_load<G>(S, st, k_lo, r));
#pragma unroll
  for (int j = 0; j < 5; j++) {
    if (j < r) {
      const int  k = k_first + DIR * j, n = n_first + DIR * j, ph = ((ph_first + DIR * j) % 3 + 3) % 3;
      const pk_t c = stage_get(st, 0, ph, k - k_lo);
      pk_t       o = 0;
      switch (ph) {
        case 0: win_step<0, MODE, W, AR>(L, v, c, 0, o); break;
        case 1: win_step<1, MODE, W, AR>(L, v, c, 0, o); break;
        default: win_step<2, MODE, W, AR>(L, v, c, 0, o); break;
      }
      if constexpr (MODE == 1) {
        if (k % CKPT == 0) beta[(k / CKPT) * 64 + L.lane] = v; // checkpoint
      }
      if ((AR || (n & 1) == 0) && n != 0) win_normalize<AR>(v);
    }
  }
}

// Elementwise phase over i = lane, lane+64, ... < n with U elements per lane in flight: every load of a batch is issued
// before its first store, so the memory round trip is paid once per batch instead of once per element. Full batches run
// without bounds checks (a guarded load is a branch, and hipcc waits for outstanding loads at every merge); the last, partial
// batch loads from clamped indices and predicates only its stores. `ld` must be a pure read.
template <int U, typename Ld, typename St>
__device__ __forceinline__ void batched(int lane, int n, Ld ld, St st)
{
  for (int base = lane; base < n; base += 64 * U) {
    decltype(ld(0)) v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = base + 64 * u;
      if (i < n) v[u] = ld(i);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = base + 64 * u;
      if (i < n) st(i, v[u]);
    }
  }
}
struct I3 { int a, b, c; };
// Eight consecutive int16 elements per lane and memory instruction (the CU issues one wavefront memory instruction per four
// cycles: the element-wise phases are bound by their number, not by bytes)
typedef short          v8s __attribute__((ext_vector_type(8)));
typedef unsigned short v8u __attribute__((ext_vector_type(8)));
typedef unsigned int   v4w __attribute__((ext_vector_type(4)));
struct V8x3 { v8s a, b; v8u c; };
struct V8W { v8s a; v4w t0, t1; };
__device__ __forceinline__ v8s ld8(const int16_t* p, int i8) { return *reinterpret_cast<const v8s*>(p + 8 * i8); }
__device__ __forceinline__ v8u ld8u(const uint16_t* p, int i8) { return *reinterpret_cast<const v8u*>(p + 8 * i8); }
__device__ __forceinline__ void st8(int16_t* p, int i8, v8s v) { *reinterpret_cast<v8s*>(p + 8 * i8) = v; }

// One SISO pass over W windows. W = 8/16 (AR = 0: sse16/avx16; AR = 1, W = 16: sse8): one pass of the wave over its 8 window
// pairs. W = 32 (avx8): the wave handles the windows as NH = 2 halves of 16, one after the other in every phase - the windows
// only meet in the two hand-overs, which see both halves' registers.
template <int W, int AR>
__device__ __forceinline__ void win_siso(const LaneGeom& L, const int16_t* __restrict__ in, const int16_t* __restrict__ app,
                         const int16_t* __restrict__ par, const int16_t* tail_in, const int16_t* tail_par, int16_t* __restrict__ out,
                         pk_t* __restrict__ beta, pk_t* __restrict__ seg, pk_t* __restrict__ scratch, const Stage& st, int K PROF_ARGS)
{
  constexpr int NH = W == 32 ? 2 : 1;
  constexpr int G  = 8 * NH; // window pairs per step
  const int     Lw = K / W;
  const pk_t    NEG = AR ? 0 : pk_make(-TD_INF, -TD_INF); // INF = 0 for the 8-bit back-ends (turbodecoder_win.h:120,:152)
  pk_t          v[NH];
  const int     top = (Lw + CKPT - 1) / CKPT; // checkpoint slot of the start metrics beta[Lw]
  const int     bstride = (top + 1) * 64;     // checkpoint columns of one half

  // ---- combine pass (turbodecoder_win.h:472-491): the four branch metrics of a step are 0, x = sat(app + syst), y = parity and
  //      x + y. Only what is not already an array in the decoder's order is written: x when there is a-priori information to
  //      add; x and y too where the inputs are not packed pairs (8 windows) or need the 8-bit representation. x + y is formed when
  //      the rows are staged into LDS (stage_store): as an array it cost one write and two reads of K values per pass.
  constexpr bool DIRECT = W != 8 && !AR;
  const int      NE     = G * Lw;
  pk_t *         Xb = scratch + NE, *Yb = scratch + 2 * NE;
  const pk_t*    Xs = DIRECT && !app ? reinterpret_cast<const pk_t*>(in) : Xb;
  constexpr bool Y_INPLACE = DIRECT || (AR && W != 8); // the parity array already is the y array (8-bit: up to the byte move of stage_store)
  const pk_t*    Ys = Y_INPLACE ? reinterpret_cast<const pk_t*>(par) : Yb;
  if constexpr (W == 8) {
    batched<8>(
        L.lane, NE, [&](int i) { return I3{ld_pair<W, AR>(in, i), app ? ld_pair<W, AR>(app, i) : 0, ld_pair<W, AR>(par, i)}; },
        [&](int i, I3 t) {
          const pk_t x = app ? s_add<AR>(t.b, t.a) : t.a;
          Xb[i]        = x;
          Yb[i]        = t.c;
        });
  } else if (!DIRECT || app) { // four window pairs (16 bytes) per lane and memory instruction; NE is a multiple of 4
    struct Q3 { int4 a, b, c; };
    auto hi8 = [](int4 v) { // AR: int8 values in int16 containers -> high bytes (see ld_pair)
      return AR ? make_int4((v.x & 0x00FF00FF) << 8, (v.y & 0x00FF00FF) << 8, (v.z & 0x00FF00FF) << 8, (v.w & 0x00FF00FF) << 8) : v;
    };
    const int4 *in4 = reinterpret_cast<const int4*>(in), *app4 = reinterpret_cast<const int4*>(app), *par4 = reinterpret_cast<const int4*>(par);
    int4 *      X4 = reinterpret_cast<int4*>(Xb), *Y4 = reinterpret_cast<int4*>(Yb);
    batched<TDEC_EWU>(
        L.lane, NE / 4,
        [&](int i) { return Q3{hi8(in4[i]), app ? hi8(app4[i]) : make_int4(0, 0, 0, 0), Y_INPLACE ? make_int4(0, 0, 0, 0) : hi8(par4[i])}; },
        [&](int i, Q3 t) {
          const int4 x = app ? make_int4(s_add<AR>(t.b.x, t.a.x), s_add<AR>(t.b.y, t.a.y), s_add<AR>(t.b.z, t.a.z), s_add<AR>(t.b.w, t.a.w)) : t.a;
          X4[i] = x;
          if (!Y_INPLACE) Y4[i] = t.c;
        });
  }
  __syncthreads();
  Src S[NH];
#pragma unroll
  for (int h = 0; h < NH; h++) S[h] = Src{Xs + h * 8, Ys + h * 8};
  PROF(2)

  // ---- beta warm-up over the first 40 steps of every window (turbodecoder_win.h:456-464); positions are fixed: all static
  static_assert(WIN_OVERLAP == 40, "block plan below is written for the 40-step overlap");
#pragma unroll
  for (int h = 0; h < NH; h++) {
    v[h]           = NEG;
    win_run<W, 24, 39 % 3, -1, 0, 1, AR>(L, v[h], S[h], st, 39, 39, 1, beta); // steps 39..16
    win_run<W, 12, 15 % 3, -1, 0, 1, AR>(L, v[h], S[h], st, 15, 15, 1, beta); // steps 15..4
    win_rem<W, -1, 0, AR>(L, v[h], S[h], st, 3, 3, 0, 4, beta);                      // steps 3..0
  }
  PROF(3)
  // ---- tail trellis for the last window: scalar (turbodecoder_win.h:351-395). 16-bit: wrapping adds; 8-bit: sadd() clamps
  //      at +127 and wraps below (:322-330), INF = 0
  int tail[8];
  {
    int o[8] = {0, -TD_INF, -TD_INF, -TD_INF, -TD_INF, -TD_INF, -TD_INF, -TD_INF};
    if constexpr (AR) {
      for (int i = 1; i < 8; i++) o[i] = 0;
    }
    auto WA = [](int a, int b) -> int {
      if constexpr (AR) {
        const int z = (int)(short)(a + b);
        return z > 127 ? 127 : (int)(signed char)z;
      } else {
        return (int)(short)(a + b);
      }
    };
    for (int j = 2; j >= 0; j--) {
      const int x = tail_in[j], y = tail_par[j], xy_ = WA(x, y);
      int m[8] = {WA(o[4], xy_), o[4], WA(o[5], y), WA(o[5], x), WA(o[6], x), WA(o[6], y), o[7], WA(o[7], xy_)};
      int n[8] = {o[0], WA(o[0], xy_), WA(o[1], x), WA(o[1], y), WA(o[2], y), WA(o[2], x), WA(o[3], xy_), o[3]};
      for (int i = 0; i < 8; i++) o[i] = m[i] > n[i] ? m[i] : n[i];
    }
    for (int i = 0; i < 8; i++) tail[i] = AR ? o[i] * 256 : o[i];
  }
  // ---- window shift: window w starts from the warm-up of window w+1, the last one from the tail (:428-448)
  {
    const int s = rotr3(L.p, Lw % 3); // state this slot holds at time Lw
    pk_t      a[NH], b[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) {
      a[h] = __shfl(v[h], lane_of(L.g, s), 64); // same group
      b[h] = __shfl(v[h], lane_of((L.g + 1) & 7, s), 64);
    }
    int ts = tail[0];
    for (int i = 1; i < 8; i++) ts = s == i ? tail[i] : ts;
    if constexpr (W == 8) {
      v[0] = pk_make(L.g == 7 ? ts : pk_lo(b[0]), 0);
    } else {
#pragma unroll
      for (int h = 0; h < NH; h++) {
        const int nxt = L.g < 7 ? pk_lo(b[h]) : (h + 1 < NH ? pk_lo(b[(h + 1) % NH]) : ts);
        v[h]          = pk_make(pk_hi(a[h]), nxt);
      }
    }
  }
  // ---- beta main pass (:466-526). Only every CKPT-th metric set is kept (in LDS, `beta`); the alpha pass recomputes the
  //      rest segment by segment. Left-over steps first, then aligned blocks.
#pragma unroll
  for (int h = 0; h < NH; h++) {
    pk_t*       bh = beta + h * bstride;
    bh[top * 64 + L.lane] = v[h];
    const int r = Lw % 6, n6 = Lw / 6, n24 = n6 / 4, r6 = n6 % 4;
    win_rem<W, -1, 1, AR>(L, v[h], S[h], st, Lw - 1, Lw - 1, (Lw - 1) % 3, r, bh);
    win_run<W, 6, 2, -1, 1, 1, AR>(L, v[h], S[h], st, Lw - r - 1, Lw - r - 1, r6, bh); // first step index = 5 (mod 6)
    win_run<W, 24, 2, -1, 1, 1, AR>(L, v[h], S[h], st, Lw - r - 1 - 6 * r6, Lw - r - 1 - 6 * r6, n24, bh);
  }

  PROF(4)
  // ---- alpha warm-up over the last 40 steps of every window (:586-603); normalisation counter j = 0..39
#pragma unroll
  for (int h = 0; h < NH; h++) {
    v[h]           = NEG;
    const int k0 = Lw - WIN_OVERLAP, ph0 = k0 % 3;
    switch (ph0) {
      case 0: win_run<W, 24, 0, 1, 0, 0, AR>(L, v[h], S[h], st, k0, 0, 1, beta); win_run<W, 12, 0, 1, 0, 0, AR>(L, v[h], S[h], st, k0 + 24, 24, 1, beta); break;
      case 1: win_run<W, 24, 1, 1, 0, 0, AR>(L, v[h], S[h], st, k0, 0, 1, beta); win_run<W, 12, 1, 1, 0, 0, AR>(L, v[h], S[h], st, k0 + 24, 24, 1, beta); break;
      default: win_run<W, 24, 2, 1, 0, 0, AR>(L, v[h], S[h], st, k0, 0, 1, beta); win_run<W, 12, 2, 1, 0, 0, AR>(L, v[h], S[h], st, k0 + 24, 24, 1, beta); break;
    }
    const int done = 36;
    win_rem<W, 1, 0, AR>(L, v[h], S[h], st, k0 + done, done, (ph0 + done) % 3, WIN_OVERLAP - done, beta);
  }
  // ---- window shift: window w starts from the end of window w-1, window 0 from the known state (:560-583)
  {
    const int sp = rotl3(L.p, Lw % 3); // slot that holds state p at time Lw
    pk_t      a[NH], b[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) {
      a[h] = __shfl(v[h], lane_of(L.g, sp), 64);
      b[h] = __shfl(v[h], lane_of((L.g + 7) & 7, sp), 64);
    }
    const int init = (AR || L.p == 0) ? 0 : -TD_INF;
    if constexpr (W == 8) {
      v[0] = pk_make(L.g == 0 ? init : pk_lo(b[0]), 0);
    } else {
#pragma unroll
      for (int h = 0; h < NH; h++) {
        const int prv = L.g > 0 ? pk_hi(b[h]) : (h > 0 ? pk_hi(b[(h + NH - 1) % NH]) : init);
        v[h]          = pk_make(prv, pk_lo(a[h]));
      }
    }
  }
  PROF(5)
  // ---- alpha main pass with extrinsic output (:605-679), CKPT steps at a time: the beta metrics of the segment are
  //      recomputed from its checkpoint into LDS (`seg`, one dword per lane and step: lane-private, no barrier), using
  //      the very operands the alpha steps need, then consumed. No beta traffic leaves the CU.
  const int nf = Lw / CKPT;
#pragma unroll
  for (int h = 0; h < NH; h++) {
    const pk_t* bh = beta + h * bstride;
    const int   gg = h * 8 + L.g;
    pk_t        va = v[h];
    StRegs sr, sr2;
    if (nf > 0) sr = sr2 = stage_load<G>(S[h], st, 0, CKPT);
    if (nf > 1) sr2 = stage_load<G>(S[h], st, CKPT, CKPT);
    for (int j = 0; j < nf; j++) {
      const int k0 = CKPT * j, buf = j & 1;
      stage_store<(W == 8 ? 0 : AR)>(st, buf, CKPT, sr);
      sr = sr2;
      if (j + 2 < nf) sr2 = stage_load<G>(S[h], st, k0 + 2 * CKPT, CKPT); // two segments of rows in flight
      pk_t c[CKPT];
#pragma unroll
      for (int i = 0; i < CKPT; i++) c[i] = st.lds[(buf ? st.rb[3 + i % 3] : st.rb[i % 3]) + i * 8]; // phase (k0 + i) % 3 = i % 3
      const pk_t Btop = bh[(j + 1) * 64 + L.lane]; // beta[k0 + CKPT], as stored (before its normalisation)
      pk_t       vb   = Btop, dummy = 0;
      if (k0 + CKPT < Lw) win_normalize<AR>(vb); // the recursion continued from the normalised value (counter != 0); beta[Lw] is a start value
#pragma unroll
      for (int i = CKPT - 1; i >= 1; i--) { // beta[k0+i], i = 23..1 (phase (k0+i)%3 = i%3)
        switch (i % 3) {
          case 0: win_step<0, 0, W, AR>(L, vb, c[i], 0, dummy); break;
          case 1: win_step<1, 0, W, AR>(L, vb, c[i], 0, dummy); break;
          default: win_step<2, 0, W, AR>(L, vb, c[i], 0, dummy); break;
        }
        seg[i * 64 + L.lane] = vb;
        if (AR || (i & 1) == 0) win_normalize<AR>(vb);
      }
#pragma unroll
      for (int i6 = 0; i6 < CKPT; i6 += 6) {
        pk_t keep = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
          const int  kk = i6 + i;
          const pk_t B  = kk == CKPT - 1 ? Btop : seg[(kk + 1) * 64 + L.lane];
          pk_t       o  = 0;
          switch (i) { // slot i keeps this step: parity i & 1
            case 0: win_step<0, 2, W, AR, 0>(L, va, c[kk], B, o); break;
            case 1: win_step<1, 2, W, AR, 1>(L, va, c[kk], B, o); break;
            case 2: win_step<2, 2, W, AR, 0>(L, va, c[kk], B, o); break;
            case 3: win_step<0, 2, W, AR, 1>(L, va, c[kk], B, o); break;
            case 4: win_step<1, 2, W, AR, 0>(L, va, c[kk], B, o); break;
            default: win_step<2, 2, W, AR, 1>(L, va, c[kk], B, o); break;
          }
          keep = L.p == i ? o : keep;
          if ((AR || (kk & 1) == 0) && (kk != 0 || k0 != 0)) win_normalize<AR>(va);
        }
        if (L.p < 6) store_out<W>(out, k0 + i6 + L.p, gg, keep);
      }
    }
    { // last, shorter segment [CKPT*nf, Lw): run-time phases, same scheme
      const int k0 = CKPT * nf, t = Lw - k0;
      if (t > 0) {
        stage_store<(W == 8 ? 0 : AR)>(st, 0, t, stage_load<G>(S[h], st, k0, t));
        const pk_t Btop = bh[top * 64 + L.lane];
        pk_t       vb   = Btop, dummy = 0;
        for (int i = t - 1; i >= 1; i--) {
          const int  ph  = (k0 + i) % 3;
          const pk_t in2 = stage_get(st, 0, ph, i);
          switch (ph) {
            case 0: win_step<0, 0, W, AR>(L, vb, in2, 0, dummy); break;
            case 1: win_step<1, 0, W, AR>(L, vb, in2, 0, dummy); break;
            default: win_step<2, 0, W, AR>(L, vb, in2, 0, dummy); break;
          }
          seg[i * 64 + L.lane] = vb;
          if (AR || ((k0 + i) & 1) == 0) win_normalize<AR>(vb);
        }
        for (int i = 0; i < t; i++) {
          const int  ph  = (k0 + i) % 3;
          const pk_t in2 = stage_get(st, 0, ph, i);
          const pk_t B   = i == t - 1 ? Btop : seg[(i + 1) * 64 + L.lane];
          pk_t       o   = 0;
          switch (ph) {
            case 0: win_step<0, 2, W, AR>(L, va, in2, B, o); break;
            case 1: win_step<1, 2, W, AR>(L, va, in2, B, o); break;
            default: win_step<2, 2, W, AR>(L, va, in2, B, o); break;
          }
          if (L.p == 0) store_out<W>(out, k0 + i, gg, o);
          if ((AR || ((k0 + i) & 1) == 0) && k0 + i != 0) win_normalize<AR>(va);
        }
      }
    }
  }
  PROF(6)
}

template <int W>
__device__ __forceinline__ int win_pos(int n, int K)
{ // natural index -> window-interleaved array position
  const int Lw = K / W;
  return (n % Lw) * W + n / Lw;
}

#include "tdec_pair.inc"

// avx8 (32 windows, 8 bit) on the pair mapping: the combine pass of win_siso<32, 1> (x = sat(a-priori + systematic) in the high bytes, to the
// xy scratch; the parity array is read in place), then the sweeps of tdec_pair.inc with the wavefront's 16 window pairs = the block's 32 windows.
// Every beta checkpoint of a window of at most 192 steps fits the LDS rows (P_CK_LDS); `ckg` only backs the clamped prefetch.
#ifndef TDEC_AR32_PAIR
#define TDEC_AR32_PAIR 1
#endif
__device__ __forceinline__ void pair_siso_ar32(const PLane& PL, int lane, const int16_t* __restrict__ in, const int16_t* __restrict__ app,
                                               const int16_t* __restrict__ par, const int16_t* tail_in, const int16_t* tail_par, int16_t* __restrict__ out,
                                               pk_t* __restrict__ pool, int2* __restrict__ ckg, pk_t* __restrict__ scratch, int K PROF_ARGS)
{
  const int NE = 16 * (K / 32);
  pk_t*     Xb = scratch + NE;
  struct Q2 { int4 a, b; };
  auto hi8 = [](int4 v) { return make_int4((v.x & 0x00FF00FF) << 8, (v.y & 0x00FF00FF) << 8, (v.z & 0x00FF00FF) << 8, (v.w & 0x00FF00FF) << 8); };
  const int4 *in4 = reinterpret_cast<const int4*>(in), *app4 = reinterpret_cast<const int4*>(app);
  int4*       X4  = reinterpret_cast<int4*>(Xb);
  batched<TDEC_EWU>(
      lane, NE / 4, [&](int i) { return Q2{hi8(in4[i]), app ? hi8(app4[i]) : make_int4(0, 0, 0, 0)}; },
      [&](int i, Q2 t) { X4[i] = app ? make_int4(s_add<1>(t.b.x, t.a.x), s_add<1>(t.b.y, t.a.y), s_add<1>(t.b.z, t.a.z), s_add<1>(t.b.w, t.a.w)) : t.a; });
  __syncthreads();
  PROF(2)
  PSrc S;
  S.x0 = Xb; S.x1 = Xb + 8; S.y0 = reinterpret_cast<const pk_t*>(par); S.y1 = S.y0 + 8;
  S.dbg = 0; S.dead0 = S.dead1 = false; S.rs = 16;
  p_siso<1, true>(PL, S, tail_in, tail_par, reinterpret_cast<pk_t*>(out), pool, ckg, K PROF_PASS);
}

// A block's contribution to its transport block's verdict (sch.c:470-488) and, from the last block to arrive, the verdict: tdec_pair_kernel's
// protocol (tdec_pair.inc, end of finish): agent-scope atomics at the memory side, the count last, after the two contributions have come back.
__device__ __forceinline__ void tdec_tb_fold(const TdecArgs& a, int tbi, int tbr, uint32_t tsyn, bool ok, bool par_nz)
{
  uint32_t*      acc = a.tb_acc + 4 * (size_t)tbi;
  const uint32_t o1  = atomicXor(acc, tsyn);
  const uint32_t o2  = atomicOr(acc + 1, (ok ? 0u : 1u) | ((tbr == (int)a.tb_C - 1 && par_nz) ? 2u : 0u));
  uint32_t       dep = o1 | o2;
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(dep)::"memory");
  if (atomicAdd(acc + 2, 1u) == a.tb_C - 1) {
    const uint32_t syn = atomicExch(acc, 0u), fl = atomicExch(acc + 1, 0u);
    atomicExch(acc + 2, 0u); // clean for the next launch
    a.tb_ok_out[tbi] = (syn == 0 && fl == 2u) ? 1 : 0;
  }
}

#ifndef TDEC_WAVES
#define TDEC_WAVES 2
#endif
#define TDEC_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(TDEC_WAVES, TDEC_WAVES)))
// AR = 1: int8 LLRs in (a.in is an int8 array), the work arrays hold int8 values in int16 containers
#ifndef TDEC_MIX_TU
template <int W, int AR>
__device__ __forceinline__ void tdec_win_body(const TdecArgs& a0, const TdecGroups& gs);
template <int W, int AR>
__global__ __launch_bounds__(64) TDEC_WAVES_ATTR void tdec_win_kernel(TdecArgs a0, TdecGroups gs)
{
  tdec_win_body<W, AR>(a0, gs);
}
#endif
// the avx8 back-end with the pair-mapped sweeps has its own register budget (see tdec_pair.inc on the 216-register rule)
#ifndef TDEC_AR32_NVGPR
#define TDEC_AR32_NVGPR 108
#endif
#if TDEC_AR32_NVGPR > 0
#define AR32_NVGPR_ATTR __attribute__((amdgpu_num_vgpr(TDEC_AR32_NVGPR)))
#else
#define AR32_NVGPR_ATTR
#endif
#ifndef TDEC_MIX_TU
__global__ __launch_bounds__(64) TDEC_WAVES_ATTR AR32_NVGPR_ATTR void tdec_ar32_kernel(TdecArgs a0, TdecGroups gs) { tdec_win_body<32, 1>(a0, gs); }
#endif
#ifdef TDEC_MIX_TU
// tdec_mix.hip: the body for a wavefront of a mixed launch - `a` already describes its group, pool / tl are the mixed kernel's LDS, laid out for
// blocks of at most WIN_MAX_K bits (AUTO gives the 8-window decoder K <= 800 only: 6 checkpoint rows instead of 34)
constexpr int WIN_MAX_K = 800;
template <int W, int AR>
__device__ __forceinline__ void tdec_win_body(TdecArgs& a, const uint32_t bx, pk_t* pool, int16_t* tl)
{
#else
constexpr int WIN_MAX_K = SRSLTE_HIP_MAX_K;
template <int W, int AR>
__device__ __forceinline__ void tdec_win_body(const TdecArgs& a0, const TdecGroups& gs)
{
  TdecArgs       a  = a0;
  const uint32_t bx = tdec_enter_group(a, gs);
#endif
  using in_t = typename std::conditional<AR != 0, int8_t, int16_t>::type;
  const int      lcb = (int)bx, cb = a.cb_map ? (int)a.cb_map[lcb] : lcb, K = (int)a.K;
  if (a.skip && a.skip[cb]) { // "Do not process blocks with CRC Ok" (sch.c:317-318): bytes, flag and TB-CRC share stay
    if (threadIdx.x == 0 && a.iters) a.iters[cb] = 0;
#ifdef TDEC_MIX_TU
    if (a.tb_out && a.tb_Cof) { // a ragged batch: the block (K <= 800) is a transport block that was delivered before; its row, no second delivery
      const uint32_t tbi = (uint32_t)cb / a.tb_C;
      const size_t   row = tbi < a.tb_B ? (size_t)tbi : (size_t)a.tb_rows0 + tbi - a.tb_B;
      for (int b = (int)threadIdx.x; b < K / 8; b += 64) a.tb_out[row * a.tb_out_stride + b] = a.out[(size_t)cb * a.out_stride + b];
      if (threadIdx.x == 0) a.tb_ok_out[row] = 0;
    }
#endif
    return;
  }
  const LaneGeom L  = lane_geom();
  const in_t*    in = reinterpret_cast<const in_t*>(a.in) + (size_t)cb * a.in_stride;
  int16_t*       wk = a.work + (size_t)lcb * 7 * a.Kp;
  int16_t *syst = wk, *par0 = wk + a.Kp, *par1 = wk + 2 * a.Kp, *app1 = wk + 3 * a.Kp, *app2 = wk + 4 * a.Kp, *ext1 = wk + 5 * a.Kp,
          *ext2 = wk + 6 * a.Kp;
  // beta checkpoints (lane-private columns): ceil(Lw / CKPT) + 1 rows per half; Lw <= MAX_K / W (two halves for W = 32)
  constexpr int BETA_ROWS = (W == 32 ? 2 : 1) * (WIN_MAX_K / W / CKPT + 2);
  // One LDS pool: beta checkpoints | beta metrics of the segment being consumed | branch-metric staging (Stage). Between SISO
  // passes the front of it (everything but the staging area's constant zero region) doubles as the buffer in which the
  // interleaver permutations are done: a 2-byte scatter costs the L1 one cache line per lane (64 cycles per wavefront
  // instruction), an LDS scatter a few bank-conflict cycles.
  constexpr int POOL_BETA = BETA_ROWS * 64, POOL_SEG = (CKPT + 1) * 64;
#ifndef TDEC_MIX_TU
  __shared__ __attribute__((aligned(16))) pk_t pool[POOL_BETA + POOL_SEG + ST_TOTAL];
#endif
  pk_t *                                       beta = pool, *seg = pool + POOL_BETA, *mt = pool + POOL_BETA + POOL_SEG;
  int16_t*                                     perm = reinterpret_cast<int16_t*>(pool);
  static_assert(2 * (POOL_BETA + POOL_SEG + 2 * ST_BUF) >= WIN_MAX_K, "permutation buffer must hold one code block");
  constexpr bool  PAIR32 = W == 32 && AR && TDEC_AR32_PAIR; // the sweeps of tdec_pair.inc; this kernel keeps extraction, exchange, CRC and decisions
  Stage st;
  if constexpr (!PAIR32) stage_init(st, mt, L);
  PROF_DECL;
  pk_t*           xy = reinterpret_cast<pk_t*>(a.xy + (size_t)lcb * a.K); // 16 B per trellis step: room for the x, y and x + y arrays
  static_assert(!PAIR32 || POOL_BETA + POOL_SEG + ST_TOTAL >= P_POOL, "the pair sweeps' LDS pool");
  const PLane     PL = p_lane<true>();
  int2*           ckg = reinterpret_cast<int2*>(a.beta) + (size_t)bx * ((K / 32 / PB + 3) * 64);
  if constexpr (PAIR32) {
    for (int i = L.lane; i < P_ARR; i += 64) pool[P_ZERO + i] = 0; // the constant zero metric, outside the permutation overlay
  }

  // ---- input extraction (turbodecoder_win.h:727-769 / turbodecoder_iter.h:58-68,84-91). The 12 tail LLRs go to LDS. A 16-bit
  //      SB-layout buffer already is three window-ordered arrays: like upstream (turbodecoder_iter.h:84-91) the decoder then
  //      reads systematic and parity LLRs in place instead of copying them.
#ifndef TDEC_MIX_TU
  __shared__ int16_t tl[12]; // [0..2] systematic tail, [3..5] parity-0 tail, [6..8] interleaved systematic tail, [9..11] parity-1 tail
#endif
  const int  tb      = a.sb_layout ? 3 * (K + 32) : 3 * K;
  const bool inplace = !AR && a.sb_layout && ((reinterpret_cast<uintptr_t>(in) | (a.in_stride * sizeof(int16_t))) & 15) == 0;
  const int16_t *syst_r = syst, *par0_r = par0, *par1_r = par1;
  if (inplace) {
    syst_r = reinterpret_cast<const int16_t*>(in);
    par0_r = syst_r + (K + 32);
    par1_r = syst_r + 2 * (K + 32);
  } else if (a.sb_layout) {
    batched<8>(
        L.lane, K, [&](int i) { return I3{in[i], in[K + 32 + i], in[2 * (K + 32) + i]}; },
        [&](int i, I3 t) {
          syst[i] = (int16_t)t.a;
          par0[i] = (int16_t)t.b;
          par1[i] = (int16_t)t.c;
        });
  } else {
    batched<8>(
        L.lane, K, [&](int n) { return I3{in[3 * n], in[3 * n + 1], in[3 * n + 2]}; },
        [&](int n, I3 t) {
          const int x = win_pos<W>(n, K);
          syst[x]     = (int16_t)t.a;
          par0[x]     = (int16_t)t.b;
          par1[x]     = (int16_t)t.c;
        });
  }
  if (L.lane < 3) {
    tl[L.lane]     = in[tb + 2 * L.lane];
    tl[3 + L.lane] = in[tb + 2 * L.lane + 1];
    tl[6 + L.lane] = in[tb + 6 + 2 * L.lane];
    tl[9 + L.lane] = in[tb + 6 + 2 * L.lane + 1];
  }
  __syncthreads();

  PROF(0)
  // srslte_vec_sub_sss: wrapping int16. srslte_vec_sub_bbb (vector_simd.c:158-185, AVX2 build, aligned buffers): saturating
  // int8 in the 32-wide body, wrapping in the scalar tail
  const int sat_body = K / 32 * 32;
  auto      vsub8    = [&](int i8, v8s x, v8s y) -> v8s { // eight elements starting at 8 * i8
    if constexpr (AR) {
      v8s r;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int d = (int)x[e] - (int)y[e];
        r[e]        = (short)(8 * i8 + e < sat_body ? max(-128, min(127, d)) : (int)(signed char)d);
      }
      return r;
    } else {
      return x - y; // wrapping, v_pk_sub_i16
    }
  };
  const int K8 = K / 8;
  constexpr int EWU = TDEC_EWU; // 16-byte elements in flight per lane in the element-wise phases
  uint32_t       n_iter = a.start_iter; // > 0: the work arrays hold the state after that many passes (tdec_set_resume)
  bool           ok     = false;
  const int16_t* dec    = ext1;
  while (n_iter < a.nof_iter && !ok) {
    uint32_t syn_fused = 0;
    bool     have_syn  = false;
    if ((n_iter & 1) == 0) {
      if (n_iter) {
        batched<EWU>(
            L.lane, K8, [&](int i8) { return V8x3{ld8(app1, i8), ld8(ext1, i8), v8u{}}; },
            [&](int i8, V8x3 t) { st8(app1, i8, vsub8(i8, t.a, t.b)); });
        __syncthreads();
      }
      PROF(1)
      if constexpr (PAIR32) {
        if (!(a.dbg & 1)) pair_siso_ar32(PL, L.lane, syst_r, n_iter ? app1 : nullptr, par0_r, tl, tl + 3, ext1, pool, ckg, xy, K PROF_PASS);
      } else {
        if (!(a.dbg & 1)) win_siso<W, AR>(L, syst_r, n_iter ? app1 : nullptr, par0_r, tl, tl + 3, ext1, beta, seg, xy, st, K PROF_PASS);
      }
      dec = ext1;
    } else {
      const bool sub = n_iter > 1 && !(a.dbg & 2); // ext1 -= app1 (srslte_vec_sub) fused with the scatter app2[deinter[i]] = ext1[i] (srslte_vec_lut)
      batched<EWU>(
          L.lane, K8, [&](int i8) { return V8x3{ld8(ext1, i8), sub ? ld8(app1, i8) : v8s{}, ld8u(a.t.deinter, i8)}; },
          [&](int i8, V8x3 t) {
            const v8s e = sub ? vsub8(i8, t.a, t.b) : t.a;
            if (sub) st8(ext1, i8, e);
#pragma unroll
            for (int j = 0; j < 8; j++) perm[t.c[j]] = e[j];
          });
      __syncthreads();
      for (int i8 = L.lane; i8 < K8; i8 += 64) st8(app2, i8, *reinterpret_cast<const v8s*>(perm + 8 * i8));
      __syncthreads();
      PROF(1)
      if constexpr (PAIR32) {
        if (!(a.dbg & 1)) pair_siso_ar32(PL, L.lane, app2, nullptr, par1_r, tl + 6, tl + 9, ext2, pool, ckg, xy, K PROF_PASS);
      } else {
        if (!(a.dbg & 1)) win_siso<W, AR>(L, app2, nullptr, par1_r, tl + 6, tl + 9, ext2, beta, seg, xy, st, K PROF_PASS);
      }
      __syncthreads();
      batched<EWU>(
          L.lane, K8, [&](int i8) { return V8x3{ld8(ext2, i8), v8s{}, ld8u(a.t.inter, i8)}; },
          [&](int i8, V8x3 t) {
#pragma unroll
            for (int j = 0; j < 8; j++) perm[t.c[j]] = t.a[j];
          });
      __syncthreads();
      if (a.t.crc_rem) { // the interleaved extrinsic values are this pass's decision metrics: their CRC syndrome (below) is taken on the way out
        batched<EWU>(
            L.lane, K8,
            [&](int i8) {
              const v4w* tp = reinterpret_cast<const v4w*>(a.t.crc_rem + 8 * i8);
              return V8W{*reinterpret_cast<const v8s*>(perm + 8 * i8), tp[0], tp[1]};
            },
            [&](int i8, V8W t) {
              st8(app1, i8, t.a);
#pragma unroll
              for (int j = 0; j < 4; j++) syn_fused ^= (t.a[j] > 0 ? t.t0[j] : 0u) ^ (t.a[4 + j] > 0 ? t.t1[j] : 0u);
            });
        have_syn = true;
      } else {
        for (int i8 = L.lane; i8 < K8; i8 += 64) st8(app1, i8, *reinterpret_cast<const v8s*>(perm + 8 * i8));
      }
      dec = app1;
    }
    __syncthreads();
    n_iter++;
    PROF(1)
    if (a.t.crc_rem && have_syn) {
      for (int o = 32; o > 0; o >>= 1) syn_fused ^= __shfl_xor(syn_fused, o, 64);
      ok = syn_fused == 0 && !(a.dbg & (1 | 16));
    } else if (a.t.crc_rem) { // sch.c:362-378: CRC over the hard decision of this pass
      uint32_t syn = 0;
      batched<EWU>(
          L.lane, K8,
          [&](int i8) {
            const v4w* tp = reinterpret_cast<const v4w*>(a.t.crc_rem + 8 * i8);
            return V8W{ld8(dec, i8), tp[0], tp[1]};
          },
          [&](int i8, V8W t) {
#pragma unroll
            for (int j = 0; j < 4; j++) syn ^= (t.a[j] > 0 ? t.t0[j] : 0u) ^ (t.a[4 + j] > 0 ? t.t1[j] : 0u);
          });
      for (int o = 32; o > 0; o >>= 1) syn ^= __shfl_xor(syn, o, 64);
      ok = syn == 0 && !(a.dbg & (1 | 16)); // 16: never stop early (timing experiments at a fixed pass count)
    }
  }
  PROF(7)
  // ---- hard decision bytes, MSB first, natural bit order (turbodecoder_win.h:771-838), and this block's share of the
  //      transport-block CRC24A syndrome (sch.c:470-488): XOR over its set payload bits of x^(position in the TB) mod g, from a table in
  //      the decoder's array order; the TB check then is an XOR of C words. One sweep over the decision metrics in array order feeds
  //      both: the syndrome directly, the bytes through LDS (consecutive bits are W elements apart in the array: as 2-byte global
  //      gathers they cost the L1 a cache line per lane).
  uint8_t*        o    = a.out + (size_t)cb * a.out_stride;
  uint32_t        tsyn = 0;
  const uint32_t* tab  = a.tb_rem ? a.tb_rem + (size_t)(cb % a.tb_C) * K : nullptr;
  __syncthreads(); // perm is free again
  if (tab) {
    batched<TDEC_EWU>(
        L.lane, K / 8,
        [&](int i8) {
          const v4w* tp = reinterpret_cast<const v4w*>(tab + 8 * i8);
          return V8W{ld8(dec, i8), tp[0], tp[1]};
        },
        [&](int i8, V8W t) {
          *reinterpret_cast<v8s*>(perm + 8 * i8) = t.a;
#pragma unroll
          for (int j = 0; j < 4; j++) tsyn ^= (t.a[j] > 0 ? t.t0[j] : 0u) ^ (t.a[4 + j] > 0 ? t.t1[j] : 0u);
        });
    for (int o2 = 32; o2 > 0; o2 >>= 1) tsyn ^= __shfl_xor(tsyn, o2, 64);
  } else {
    for (int i8 = L.lane; i8 < K8; i8 += 64) *reinterpret_cast<v8s*>(perm + 8 * i8) = ld8(dec, i8);
  }
  __syncthreads();
  // transport-block assembly by the decoder itself (tdec_set_tb_direct, as tdec_pair_kernel's last phase; sch.c:360,:401-410,:470-488): block r of
  // transport block cb / C owns bytes [r rb, r rb + rb) of it, the last block also leaves its own CRC behind the TB's
  const int tbr = (int)(cb % a.tb_C), tbi = (int)(cb / a.tb_C);
  int       rb = (int)a.tb_rb, tbC = (int)a.tb_C;
  size_t    row = (size_t)tbi;
#ifdef TDEC_MIX_TU
  if (a.tb_Cof) { // a ragged batch: the blocks this decoder takes (K <= 800) are transport blocks of their own, their CRC24A is the block's CRC
    tbC = 1;
    rb  = K8;
    row = (uint32_t)tbi < a.tb_B ? (size_t)tbi : (size_t)a.tb_rows0 + tbi - a.tb_B;
  }
#endif
  uint8_t*  tbo = a.tb_out ? a.tb_out + row * a.tb_out_stride + (size_t)tbr * rb : nullptr;
  const int tlim = tbr == tbC - 1 ? K8 : rb;
  bool      par_nz = false; // the TB's CRC24A bytes, the last three of the last block's rb: not all zero (sch.c:481)
  {
    const int      Lw    = K / W;
    const uint32_t magic = (uint32_t)((0x100000000ull + (uint32_t)Lw - 1) / (uint32_t)Lw); // n / Lw = (n * magic) >> 32 for n < 2^32 / Lw
    for (int b = L.lane; b < K8; b += 64) {
      int      q = (int)__umulhi((uint32_t)(8 * b), magic), r = 8 * b - q * Lw; // natural index n -> array position (n % Lw) * W + n / Lw
      uint32_t byte = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        byte |= (perm[r * W + q] > 0 ? 0x80u : 0u) >> j;
        if (++r == Lw) {
          r = 0;
          q++;
        }
      }
      o[b] = (uint8_t)byte;
      if (tbo && b < tlim) tbo[b] = (uint8_t)byte;
      par_nz = par_nz || (b >= rb - 3 && b < rb && byte != 0);
    }
  }
  if (tbo) par_nz = __ballot(par_nz) != 0;
  PROF(8)
  if (L.lane == 0) {
#ifdef TDEC_MIX_TU
    if (tbo && a.tb_Cof) a.tb_ok_out[row] = (ok && par_nz) ? 1 : 0; // one block: the verdict of sch.c:470-488 is the block's CRC and a non-zero parity
    else
#endif
    if (tbo) tdec_tb_fold(a, tbi, tbr, tsyn, ok, par_nz);
    if (a.iters) a.iters[cb] = n_iter;
    if (a.crc_ok) a.crc_ok[cb] = ok ? 1 : 0;
    if (a.tb_rem) a.tb_syn[cb] = tsyn;
#ifdef TDEC_PROF
    if (a.prof) {
      for (int i = 0; i < 10; i++) a.prof[(size_t)cb * 10 + i] = prof_acc[i];
    }
#endif
  }
}


// ------------------------------------------------------------------------------------------------------------------
// sse8 (16 windows, 8 bit: 800 < K <= 2048 under the 8-bit API) on the pair mapping: TWO code blocks per wavefront, as tdec_pair_kernel
// runs the 16-bit blocks - lanes 0-31 sweep block slot 0, lanes 32-63 slot 1, eight window pairs each - with the 8-bit arithmetic of the
// sweeps' AR template (turbodecoder_win.h:92-173). Everything around the sweeps is tdec_win_body<16, 1>'s, slot after slot: extraction,
// srslte_vec_sub_bbb's order of operations (saturating in the 32-wide body, wrapping in the scalar tail, vector_simd.c:158-185), the two
// permutations through LDS, the CRC syndrome of every pass, decisions. A slot whose block has passed its CRC (or that has no block: an odd
// count, a skipped block) waits in lockstep: its lanes read its partner's first rows over and over and store nothing.
// ------------------------------------------------------------------------------------------------------------------
#ifndef TDEC_AR16_PAIR
#define TDEC_AR16_PAIR 1 // 0: the state-per-lane kernel of rounds 1-3 (tdec_win_kernel<16, 1>), for A/B builds
#endif
#ifndef TDEC_AR16_NVGPR
#define TDEC_AR16_NVGPR 108
#endif
#if TDEC_AR16_NVGPR > 0
#define AR16_NVGPR_ATTR __attribute__((amdgpu_num_vgpr(TDEC_AR16_NVGPR)))
#else
#define AR16_NVGPR_ATTR
#endif
#ifndef TDEC_MIX_TU
__global__ __launch_bounds__(64) TDEC_WAVES_ATTR AR16_NVGPR_ATTR void tdec_ar16_kernel(TdecArgs a0, TdecGroups gs)
{
  TdecArgs       a  = a0;
  const uint32_t bx = tdec_enter_group(a, gs);
  const int      K = (int)a.K, K8 = K / 8, Kp = (int)a.Kp;
  struct Blk {
    int  cb, lcb;
    bool ok; // passed its CRC, or nothing to decode in this slot
    bool live;
    uint32_t its;
    const int16_t* dec;
  } B[2];
  for (int s_ = 0; s_ < 2; s_++) {
    const int lcb = 2 * (int)bx + s_;
    B[s_].live = lcb < (int)a.nof_cb;
    B[s_].lcb  = B[s_].live ? lcb : 2 * (int)bx;
    B[s_].cb   = a.cb_map ? (int)a.cb_map[B[s_].lcb] : B[s_].lcb;
    if (B[s_].live && a.skip && a.skip[B[s_].cb]) { // "Do not process blocks with CRC Ok" (sch.c:317-318): bytes, flag and TB-CRC share stay
      if (threadIdx.x == 0 && a.iters) a.iters[B[s_].cb] = 0;
      B[s_].live = false;
    }
    B[s_].ok  = !B[s_].live;
    B[s_].its = a.start_iter;
    B[s_].dec = nullptr;
  }
  if (!B[0].live && !B[1].live) return;
  const int      lane = threadIdx.x & 63;
  const PLane    PL   = p_lane<false>();
  __shared__ __attribute__((aligned(16))) pk_t pool[P_POOL];
  __shared__ int16_t tl[2][12]; // per slot: [0..2] systematic tail, [3..5] parity-0 tail, [6..8] interleaved systematic tail, [9..11] parity-1 tail
  int16_t*       perm = reinterpret_cast<int16_t*>(pool);
  static_assert(2 * P_OVER >= 2048, "permutation buffer must hold one sse8 code block");
  for (int i = lane; i < P_ARR; i += 64) pool[P_ZERO + i] = 0; // the constant zero metric, outside the permutation overlay
  PROF_DECL;
  int2*          ckg = reinterpret_cast<int2*>(a.beta) + (size_t)bx * ((K / 16 / PB + 3) * 64);
  auto WKA = [&](const Blk& b, int i) { return a.work + (size_t)b.lcb * 7 * Kp + (size_t)i * Kp; }; // 0 syst 1 par0 2 par1 3 app1 4 app2 5 ext1 6 ext2
  auto XB  = [&](const Blk& b) { return reinterpret_cast<pk_t*>(a.xy + (size_t)b.lcb * a.K) + K / 2; };     // the combined x rows of the slot's block

  // ---- input extraction (turbodecoder_win.h:727-769): int8 LLRs to int16 containers in window order, the 12 tail LLRs to LDS
  const int tb = a.sb_layout ? 3 * (K + 32) : 3 * K;
  for (int s_ = 0; s_ < 2; s_++) {
    if (!B[s_].live) continue;
    const int8_t* in = reinterpret_cast<const int8_t*>(a.in) + (size_t)B[s_].cb * a.in_stride;
    int16_t *syst = WKA(B[s_], 0), *par0 = WKA(B[s_], 1), *par1 = WKA(B[s_], 2);
    if (a.sb_layout) {
      batched<8>(
          lane, K, [&](int i) { return I3{in[i], in[K + 32 + i], in[2 * (K + 32) + i]}; },
          [&](int i, I3 t) {
            syst[i] = (int16_t)t.a;
            par0[i] = (int16_t)t.b;
            par1[i] = (int16_t)t.c;
          });
    } else {
      batched<8>(
          lane, K, [&](int n) { return I3{in[3 * n], in[3 * n + 1], in[3 * n + 2]}; },
          [&](int n, I3 t) {
            const int x = win_pos<16>(n, K);
            syst[x]     = (int16_t)t.a;
            par0[x]     = (int16_t)t.b;
            par1[x]     = (int16_t)t.c;
          });
    }
    if (lane < 3) {
      tl[s_][lane]     = in[tb + 2 * lane];
      tl[s_][3 + lane] = in[tb + 2 * lane + 1];
      tl[s_][6 + lane] = in[tb + 6 + 2 * lane];
      tl[s_][9 + lane] = in[tb + 6 + 2 * lane + 1];
    }
  }
  if (!B[0].live || !B[1].live) { // the empty slot's lanes read the other slot's tails (their results go nowhere)
    const int from = B[0].live ? 0 : 1;
    __syncthreads();
    if (lane < 12) tl[1 - from][lane] = tl[from][lane];
  }
  __syncthreads();
  PROF(0)
  const int sat_body = K / 32 * 32; // srslte_vec_sub_bbb: saturating int8 in the 32-wide body, wrapping in the scalar tail
  auto      vsub8    = [&](int i8, v8s x, v8s y) -> v8s {
    v8s r;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int d = (int)x[e] - (int)y[e];
      r[e]        = (short)(8 * i8 + e < sat_body ? max(-128, min(127, d)) : (int)(signed char)d);
    }
    return r;
  };
  constexpr int EWU = TDEC_EWU;
  // x = sat(a-priori + systematic) in the high bytes, to the slot's scratch rows (the combine pass of win_siso<16, 1>)
  auto combine = [&](const Blk& b, const int16_t* in, const int16_t* app) {
    struct Q2 { int4 a, b; };
    auto hi8 = [](int4 v) { return make_int4((v.x & 0x00FF00FF) << 8, (v.y & 0x00FF00FF) << 8, (v.z & 0x00FF00FF) << 8, (v.w & 0x00FF00FF) << 8); };
    const int4 *in4 = reinterpret_cast<const int4*>(in), *app4 = reinterpret_cast<const int4*>(app);
    int4*       X4  = reinterpret_cast<int4*>(XB(b));
    batched<EWU>(
        lane, K / 8, [&](int i) { return Q2{hi8(in4[i]), app ? hi8(app4[i]) : make_int4(0, 0, 0, 0)}; },
        [&](int i, Q2 t) { X4[i] = app ? make_int4(s_add<1>(t.b.x, t.a.x), s_add<1>(t.b.y, t.a.y), s_add<1>(t.b.z, t.a.z), s_add<1>(t.b.w, t.a.w)) : t.a; });
  };
  uint32_t n_iter = a.start_iter; // > 0: the work arrays hold the state after that many passes (tdec_set_resume)
  while (n_iter < a.nof_iter && !(B[0].ok && B[1].ok)) {
    const bool odd = (n_iter & 1) != 0;
    for (int s_ = 0; s_ < 2; s_++) {
      Blk& b = B[s_];
      if (b.ok) continue;
      int16_t *syst = WKA(b, 0), *app1 = WKA(b, 3), *app2 = WKA(b, 4), *ext1 = WKA(b, 5);
      if (!odd) {
        if (n_iter) {
          batched<EWU>(
              lane, K8, [&](int i8) { return V8x3{ld8(app1, i8), ld8(ext1, i8), v8u{}}; },
              [&](int i8, V8x3 t) { st8(app1, i8, vsub8(i8, t.a, t.b)); });
          __syncthreads();
        }
        combine(b, syst, n_iter ? app1 : nullptr);
      } else {
        const bool sub = n_iter > 1; // ext1 -= app1 (srslte_vec_sub) fused with the scatter app2[deinter[i]] = ext1[i] (srslte_vec_lut)
        batched<EWU>(
            lane, K8, [&](int i8) { return V8x3{ld8(ext1, i8), sub ? ld8(app1, i8) : v8s{}, ld8u(a.t.deinter, i8)}; },
            [&](int i8, V8x3 t) {
              const v8s e = sub ? vsub8(i8, t.a, t.b) : t.a;
              if (sub) st8(ext1, i8, e);
#pragma unroll
              for (int j = 0; j < 8; j++) perm[t.c[j]] = e[j];
            });
        __syncthreads();
        for (int i8 = lane; i8 < K8; i8 += 64) st8(app2, i8, *reinterpret_cast<const v8s*>(perm + 8 * i8));
        __syncthreads();
        combine(b, app2, nullptr);
      }
    }
    __syncthreads();
    PROF(1)
    {
      const Blk& f = B[0].ok ? B[1] : B[0]; // a stopped slot reads valid memory: the running one's rows
      auto       src = [&](int s_) -> const Blk& { return B[s_].ok ? f : B[s_]; };
      PSrc       S;
      S.x0 = XB(src(0)); S.x1 = XB(src(1));
      S.y0 = reinterpret_cast<const pk_t*>(WKA(src(0), odd ? 2 : 1)); S.y1 = reinterpret_cast<const pk_t*>(WKA(src(1), odd ? 2 : 1));
      S.dbg = 0; S.dead0 = B[0].ok; S.dead1 = B[1].ok; S.rs = 8;
      pk_t* out = reinterpret_cast<pk_t*>(WKA(src(PL.h), odd ? 6 : 5)); // a dead slot's lanes store nothing (p_siso: wr)
      p_siso<1, false>(PL, S, tl[PL.h] + (odd ? 6 : 0), tl[PL.h] + (odd ? 9 : 3), out, pool, ckg, K PROF_PASS);
    }
    __syncthreads();
    n_iter++;
    for (int s_ = 0; s_ < 2; s_++) {
      Blk& b = B[s_];
      if (b.ok) continue;
      b.its = n_iter;
      int16_t *app1 = WKA(b, 3), *ext1 = WKA(b, 5), *ext2 = WKA(b, 6);
      uint32_t syn = 0;
      if (!odd) {
        b.dec = ext1;
        if (a.t.crc_rem) { // sch.c:362-378: CRC over the hard decision of this pass
          batched<EWU>(
              lane, K8,
              [&](int i8) {
                const v4w* tp = reinterpret_cast<const v4w*>(a.t.crc_rem + 8 * i8);
                return V8W{ld8(ext1, i8), tp[0], tp[1]};
              },
              [&](int i8, V8W t) {
#pragma unroll
                for (int j = 0; j < 4; j++) syn ^= (t.a[j] > 0 ? t.t0[j] : 0u) ^ (t.a[4 + j] > 0 ? t.t1[j] : 0u);
              });
        }
      } else {
        batched<EWU>(
            lane, K8, [&](int i8) { return V8x3{ld8(ext2, i8), v8s{}, ld8u(a.t.inter, i8)}; },
            [&](int i8, V8x3 t) {
#pragma unroll
              for (int j = 0; j < 8; j++) perm[t.c[j]] = t.a[j];
            });
        __syncthreads();
        if (a.t.crc_rem) { // the interleaved extrinsic values are this pass's decision metrics: their CRC syndrome is taken on the way out
          batched<EWU>(
              lane, K8,
              [&](int i8) {
                const v4w* tp = reinterpret_cast<const v4w*>(a.t.crc_rem + 8 * i8);
                return V8W{*reinterpret_cast<const v8s*>(perm + 8 * i8), tp[0], tp[1]};
              },
              [&](int i8, V8W t) {
                st8(app1, i8, t.a);
#pragma unroll
                for (int j = 0; j < 4; j++) syn ^= (t.a[j] > 0 ? t.t0[j] : 0u) ^ (t.a[4 + j] > 0 ? t.t1[j] : 0u);
              });
        } else {
          for (int i8 = lane; i8 < K8; i8 += 64) st8(app1, i8, *reinterpret_cast<const v8s*>(perm + 8 * i8));
        }
        b.dec = app1;
        __syncthreads(); // perm is free again, app1 written
      }
      if (a.t.crc_rem) {
        for (int o = 32; o > 0; o >>= 1) syn ^= __shfl_xor(syn, o, 64);
        b.ok = syn == 0;
      }
    }
    __syncthreads();
    PROF(7)
  }
  // ---- hard decision bytes, MSB first, natural bit order (turbodecoder_win.h:771-838), and the block's share of the transport-block
  //      CRC24A syndrome (sch.c:470-488), slot after slot as tdec_win_body does for its one block
  for (int s_ = 0; s_ < 2; s_++) {
    const Blk& b = B[s_];
    if (!b.live) continue;
    const int16_t*  dec  = b.dec ? b.dec : WKA(b, (a.start_iter & 1) ? 3 : 5); // resumed at the pass limit: the kept state's decision metrics
    uint8_t*        o    = a.out + (size_t)b.cb * a.out_stride;
    uint32_t        tsyn = 0;
    const uint32_t* tab  = a.tb_rem ? a.tb_rem + (size_t)(b.cb % a.tb_C) * K : nullptr;
    __syncthreads(); // perm is free again
    if (tab) {
      batched<TDEC_EWU>(
          lane, K8,
          [&](int i8) {
            const v4w* tp = reinterpret_cast<const v4w*>(tab + 8 * i8);
            return V8W{ld8(dec, i8), tp[0], tp[1]};
          },
          [&](int i8, V8W t) {
            *reinterpret_cast<v8s*>(perm + 8 * i8) = t.a;
#pragma unroll
            for (int j = 0; j < 4; j++) tsyn ^= (t.a[j] > 0 ? t.t0[j] : 0u) ^ (t.a[4 + j] > 0 ? t.t1[j] : 0u);
          });
      for (int o2 = 32; o2 > 0; o2 >>= 1) tsyn ^= __shfl_xor(tsyn, o2, 64);
    } else {
      for (int i8 = lane; i8 < K8; i8 += 64) *reinterpret_cast<v8s*>(perm + 8 * i8) = ld8(dec, i8);
    }
    __syncthreads();
    const int tbr = (int)(b.cb % a.tb_C), tbi = (int)(b.cb / a.tb_C), rb = (int)a.tb_rb; // transport-block assembly, as tdec_win_body
    uint8_t*  tbo = a.tb_out ? a.tb_out + (size_t)tbi * a.tb_out_stride + (size_t)tbr * rb : nullptr;
    const int tlim = tbr == (int)a.tb_C - 1 ? K8 : rb;
    bool      par_nz = false;
    {
      const int      Lw    = K / 16;
      const uint32_t magic = (uint32_t)((0x100000000ull + (uint32_t)Lw - 1) / (uint32_t)Lw); // n / Lw = (n * magic) >> 32 for n < 2^32 / Lw
      for (int bb = lane; bb < K8; bb += 64) {
        int      q = (int)__umulhi((uint32_t)(8 * bb), magic), r = 8 * bb - q * Lw; // natural index n -> array position (n % Lw) * 16 + n / Lw
        uint32_t byte = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          byte |= (perm[r * 16 + q] > 0 ? 0x80u : 0u) >> j;
          if (++r == Lw) {
            r = 0;
            q++;
          }
        }
        o[bb] = (uint8_t)byte;
        if (tbo && bb < tlim) tbo[bb] = (uint8_t)byte;
        par_nz = par_nz || (bb >= rb - 3 && bb < rb && byte != 0);
      }
    }
    if (tbo) par_nz = __ballot(par_nz) != 0;
    if (lane == 0) {
      if (tbo) tdec_tb_fold(a, tbi, tbr, tsyn, b.ok && a.t.crc_rem, par_nz);
      if (a.iters) a.iters[b.cb] = b.its;
      if (a.crc_ok) a.crc_ok[b.cb] = (b.ok && a.t.crc_rem) ? 1 : 0;
      if (a.tb_rem) a.tb_syn[b.cb] = tsyn;
    }
  }
  PROF(8)
}


#endif // !TDEC_MIX_TU

// ------------------------------------------------------------------------------------------------------------------
// Generic decoder (turbodecoder_gen.c:54-233): 8 code blocks per wave (group g = block slot), low half only, wrapping
// ------------------------------------------------------------------------------------------------------------------
// The two recursions are chains of K + 3 / K dependent steps; what a step NEEDS from memory (systematic, a-priori and parity value of its
// index, and the beta row in the alpha pass) does not depend on the chain. GEN_CH steps' worth is requested at once, ahead of the
// steps that use it: a step per load round trip (230 cycles, the round-2 form) becomes twelve (profiles/r04/ab_gen_prefetch.txt). The
// arithmetic of every step is unchanged.
constexpr int GEN_CH = 12; // multiple of 3 (trellis phases) and of 4 (normalisation period)
__device__ void gen_siso(const LaneGeom& L, const int16_t* __restrict__ in, const int16_t* __restrict__ app,
                         const int16_t* __restrict__ par, int16_t* __restrict__ out, pk_t* __restrict__ beta, int K, bool active)
{
  constexpr bool SAT = false;
  pk_t           v, to, tp;
  const int      end = K + 3;
#define ACS(ph)                                                        \
  switch (ph) {                                                        \
    case 0: v = acs<0, SAT>(L, v, x, y, xy, &to, &tp); break;          \
    case 1: v = acs<1, SAT>(L, v, x, y, xy, &to, &tp); break;          \
    default: v = acs<2, SAT>(L, v, x, y, xy, &to, &tp); break;         \
  }
  // beta (:54-110): known end state after the 3 tail steps
  v                        = pk_make(L.p == 0 ? 0 : -TD_INF, 0);
  beta[end * 64 + L.lane] = v;
  int k = end - 1;
  for (int ph = k % 3; (k + 1) % GEN_CH != 0; k--, ph = ph ? ph - 1 : 2) { // the steps above the last multiple of GEN_CH: the tail among them
    int xi = in[k];
    if (app && k < K) xi += app[k];
    const pk_t x = pk_make(xi, 0), y = pk_make(par[k], 0), xy = pk_add<SAT>(x, y);
    ACS(ph);
    beta[k * 64 + L.lane] = v;
    if ((k & 3) == 0 && k < K) v = pk_sub<SAT>(v, bcast_slot0(v));
  }
  for (; k >= 0; k -= GEN_CH) { // k = GEN_CH - 1 (mod GEN_CH): phases 2, 1, 0, ..., normalisation where (k - i) % 4 == 0
    int xs[GEN_CH], as[GEN_CH], ys[GEN_CH];
#pragma unroll
    for (int i = 0; i < GEN_CH; i++) {
      xs[i] = in[k - i];
      as[i] = app ? app[k - i] : 0;
      ys[i] = par[k - i];
    }
#pragma unroll
    for (int i = 0; i < GEN_CH; i++) {
      const int  kk = k - i;
      const int  xi = xs[i] + ((app && kk < K) ? as[i] : 0);
      const pk_t x = pk_make(xi, 0), y = pk_make(ys[i], 0), xy = pk_add<SAT>(x, y);
      if (i % 3 == 0) v = acs<2, SAT>(L, v, x, y, xy, &to, &tp);
      else if (i % 3 == 1) v = acs<1, SAT>(L, v, x, y, xy, &to, &tp);
      else v = acs<0, SAT>(L, v, x, y, xy, &to, &tp);
      beta[kk * 64 + L.lane] = v;
      if ((GEN_CH - 1 - i) % 4 == 0 && kk < K) v = pk_sub<SAT>(v, bcast_slot0(v));
    }
  }
  __syncthreads();
  // alpha (:112-194)
  v = pk_make(L.p == 0 ? 0 : -TD_INF, 0);
  auto alpha_step = [&](int kk, int ph, int xi, int yv, pk_t B) {
    const pk_t x = pk_make(xi, 0), y = pk_make(yv, 0), xy = pk_add<SAT>(x, y);
    ACS(ph);
    const bool b0 = ph == 0 ? L.p1 : (ph == 1 ? L.p2 : L.p0);
    pk_t       m0 = group_max(pk_add<SAT>(B, b0 ? tp : to));
    pk_t       m1 = group_max(pk_add<SAT>(B, b0 ? to : tp));
    if ((kk & 3) == 0) v = pk_sub<SAT>(v, bcast_slot0(v));
    if (L.p == 0 && active) out[kk - 1] = (int16_t)pk_lo(pk_sub<SAT>(m1, m0));
  };
  k = 1;
  for (; k + GEN_CH <= K + 1; k += GEN_CH) { // k = 1 (mod GEN_CH): phases 0, 1, 2, ..., normalisation where (1 + i) % 4 == 0
    int  xs[GEN_CH], as[GEN_CH], ys[GEN_CH];
    pk_t Bs[GEN_CH];
#pragma unroll
    for (int i = 0; i < GEN_CH; i++) {
      xs[i] = in[k - 1 + i];
      as[i] = app ? app[k - 1 + i] : 0;
      ys[i] = par[k - 1 + i];
      Bs[i] = beta[(k + i) * 64 + L.lane];
    }
#pragma unroll
    for (int i = 0; i < GEN_CH; i++) alpha_step(k + i, i % 3, xs[i] + as[i], ys[i], Bs[i]);
  }
  for (int ph = (k - 1) % 3; k < K + 1; k++, ph = ph == 2 ? 0 : ph + 1) {
    int xi = in[k - 1];
    if (app) xi += app[k - 1];
    alpha_step(k, ph, xi, par[k - 1], beta[k * 64 + L.lane]);
  }
#undef ACS
}

#ifdef TDEC_MIX_TU
__device__ __forceinline__ void tdec_gen_body(TdecArgs& a, const uint32_t bx)
{
#else
__global__ __launch_bounds__(64) void tdec_gen_kernel(TdecArgs a0, TdecGroups gs)
{
  TdecArgs       a  = a0;
  const uint32_t bx = tdec_enter_group(a, gs);
#endif
  const LaneGeom L      = lane_geom();
  const int      K      = (int)a.K;
  const uint32_t cb_raw = bx * 8 + L.g;
  const uint32_t lcb    = cb_raw < a.nof_cb ? cb_raw : a.nof_cb - 1; // idle groups shadow the last block, stores predicated
  const uint32_t cb     = a.cb_map ? a.cb_map[lcb] : lcb;
  const bool     skipped = cb_raw < a.nof_cb && a.skip && a.skip[cb]; // sch.c:317-318
  const bool     active = cb_raw < a.nof_cb && !skipped;
  const int16_t* in     = a.in + (size_t)cb * a.in_stride;
  int16_t*       wk     = a.work + (size_t)lcb * 7 * a.Kp;
  int16_t *syst = wk, *par0 = wk + a.Kp, *par1 = wk + 2 * a.Kp, *app1 = wk + 3 * a.Kp, *app2 = wk + 4 * a.Kp, *ext1 = wk + 5 * a.Kp,
          *ext2 = wk + 6 * a.Kp;
  pk_t* beta = a.beta + (size_t)bx * a.beta_stride;

  // extraction (turbodecoder_gen.c:235-253): each group's 8 slots stride over their block
  if (active) {
    for (int n = L.p; n < K; n += 8) {
      syst[n] = in[3 * n];
      par0[n] = in[3 * n + 1];
      par1[n] = in[3 * n + 2];
    }
    if (L.p < 3) {
      syst[K + L.p] = in[3 * K + 2 * L.p];
      par0[K + L.p] = in[3 * K + 2 * L.p + 1];
      app2[K + L.p] = in[3 * K + 6 + 2 * L.p];
      par1[K + L.p] = in[3 * K + 6 + 2 * L.p + 1];
    }
  }
  __syncthreads();

  // all 8 blocks of a wave run the same number of passes; a block whose CRC already passed keeps its result
  uint32_t       my_iters = 0;
  bool           ok       = false;
  bool           use_app1 = false;
  const uint32_t nbytes   = K / 8;
  uint8_t*       o        = a.out + (size_t)cb * a.out_stride;
  for (uint32_t n_iter = a.start_iter; n_iter < a.nof_iter; n_iter++) {
    const bool run = active && !ok;
    if ((n_iter & 1) == 0) {
      if (n_iter && run) {
        for (int i = L.p; i < K; i += 8) app1[i] = (int16_t)(app1[i] - ext1[i]);
      }
      __syncthreads();
      gen_siso(L, syst, n_iter ? app1 : nullptr, par0, ext1, beta, K, run);
      use_app1 = false;
    } else {
      if (run) {
        if (n_iter > 1) {
          for (int i = L.p; i < K; i += 8) ext1[i] = (int16_t)(ext1[i] - app1[i]);
        }
        __syncthreads();
        for (int i = L.p; i < K; i += 8) app2[a.t.deinter[i]] = ext1[i];
      } else {
        __syncthreads();
      }
      __syncthreads();
      gen_siso(L, app2, nullptr, par1, ext2, beta, K, run);
      __syncthreads();
      if (run) {
        for (int i = L.p; i < K; i += 8) app1[a.t.inter[i]] = ext2[i];
      }
      use_app1 = true;
    }
    __syncthreads();
    if (run) {
      my_iters            = n_iter + 1;
      const int16_t* dec  = use_app1 ? app1 : ext1;
      uint32_t       syn  = 0;
      for (uint32_t b = L.p; b < nbytes; b += 8) { // hard decision of this pass (turbodecoder_gen.c:255-273)
        uint32_t byte = 0;
        for (int j = 0; j < 8; j++) {
          const bool bit = dec[8 * b + j] > 0;
          byte |= (bit ? 0x80u : 0u) >> j;
          if (a.t.crc_rem && bit) syn ^= a.t.crc_rem[8 * b + j];
        }
        o[b] = (uint8_t)byte;
      }
      if (a.t.crc_rem) {
        syn ^= __builtin_amdgcn_update_dpp(syn, syn, 0xB1, 0xf, 0xf, false);
        syn ^= __builtin_amdgcn_update_dpp(syn, syn, 0x4E, 0xf, 0xf, false);
        syn ^= __builtin_amdgcn_update_dpp(syn, syn, 0x128, 0xf, 0xf, false);
        ok = syn == 0;
      }
    }
    if (__all(ok || !active)) break; // sch.c:383 per block; the wave leaves when every block has stopped
  }
#ifdef TDEC_MIX_TU
  if (a.tb_out && a.tb_Cof && active) { // a ragged batch: the block is a transport block of its own (K <= 400): its bytes and its verdict
    const uint32_t tbi = cb / a.tb_C;
    const size_t   row = tbi < a.tb_B ? (size_t)tbi : (size_t)a.tb_rows0 + tbi - a.tb_B;
    uint8_t*       tbo = a.tb_out + row * a.tb_out_stride;
    bool           par_nz = false;
    for (uint32_t b = L.p; b < nbytes; b += 8) { // the bytes this lane wrote in the block's last pass
      const uint8_t v = o[b];
      tbo[b]          = v;
      par_nz          = par_nz || (b + 3 >= nbytes && v != 0);
    }
    uint32_t pz = par_nz ? 1u : 0u;
    pz |= __builtin_amdgcn_update_dpp(pz, pz, 0xB1, 0xf, 0xf, false);
    pz |= __builtin_amdgcn_update_dpp(pz, pz, 0x4E, 0xf, 0xf, false);
    pz |= __builtin_amdgcn_update_dpp(pz, pz, 0x128, 0xf, 0xf, false);
    if (L.p == 0) a.tb_ok_out[row] = (ok && pz) ? 1 : 0;
  }
#endif
  if (L.p == 0 && active) {
    if (a.iters) a.iters[cb] = my_iters;
    if (a.crc_ok) a.crc_ok[cb] = ok ? 1 : 0;
  }
  if (L.p == 0 && skipped && a.iters) a.iters[cb] = 0;
#ifdef TDEC_MIX_TU
  if (a.tb_out && a.tb_Cof && skipped) { // delivered by an earlier transmission: the stored bytes, no second delivery
    const uint32_t tbi = cb / a.tb_C;
    const size_t   row = tbi < a.tb_B ? (size_t)tbi : (size_t)a.tb_rows0 + tbi - a.tb_B;
    for (uint32_t b = L.p; b < (uint32_t)K / 8; b += 8) a.tb_out[row * a.tb_out_stride + b] = a.out[(size_t)cb * a.out_stride + b];
    if (L.p == 0) a.tb_ok_out[row] = 0;
  }
#endif
}

#ifdef TDEC_MIX_TU
// ------------------------------------------------------------------------------------------------------------------
// A ragged batch (tdec_run_groups) in ONE launch: its groups of equal block length may be of different KINDS - the two-blocks-per-wavefront
// decoder (K > 800), the 8-window one (400 < K <= 800), the unwindowed one (K <= 400). Launched one kind after the other on the batch's stream
// they were a chain: the short-block launches are a few wavefronts each and as long as their longest recursion (unwindowed 127 us, 8-window
// 59 us behind the 376 us of the long blocks in the mixed-grant workload). Here every wavefront looks up its group, reads its kind and runs
// that decoder's body; the LDS pool is laid out for the larger need (pair sweeps / the 8-window body with the checkpoint rows of K <= 800).
// Register budget and occupancy: the pair kernel's. Mixed-grant line 622 k -> 791 k subframes/s (profiles/r04/ab_mix_kernel.txt).
// ------------------------------------------------------------------------------------------------------------------
constexpr int MIX_WIN8_POOL = (WIN_MAX_K / 8 / CKPT + 2) * 64 + (CKPT + 1) * 64 + ST_TOTAL;
constexpr int MIX_POOL      = MIX_WIN8_POOL > P_POOL ? MIX_WIN8_POOL : P_POOL;
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TDEC_PAIR_MINW, TDEC_PAIR_WAVES))) P_NVGPR_ATTR void tdec_mix_kernel(TdecArgs a0, TdecGroups gs)
{
  TdecArgs       a = a0;
  uint32_t       gi = 0;
  for (uint32_t i = 1; i < gs.n; i++) gi = blockIdx.x >= gs.g[i].first_wave ? i : gi;
  const uint32_t kind = gs.kind[gi];
  const uint32_t bx   = tdec_enter_group(a, gs);
  __shared__ __attribute__((aligned(16))) pk_t pool[MIX_POOL];
  __shared__ int16_t tl[2][12];
  a.sb_layout = kind != TDEC_KIND_GEN; // the unwindowed decoder reads the plain [s p0 p1] layout, the windowed ones the rate de-matcher's
  a.tbA       = gs.tbA[gi];
  if (kind == TDEC_KIND_PAIR) tdec_pair_body(a, bx, pool, tl);
  else if (kind == TDEC_KIND_WIN8) tdec_win_body<8, 0>(a, bx, pool, tl[0]);
  else tdec_gen_body(a, bx);
}
} // namespace
// called by tdec_run_groups (tdec.hip); args / groups: that unit's TdecArgs / TdecGroups (the same definitions)
int tdec_mix_launch(const void* args, const void* groups, unsigned waves, hipStream_t st)
{
  hipLaunchKernelGGL(tdec_mix_kernel, dim3(waves), dim3(64), 0, st, *static_cast<const TdecArgs*>(args), *static_cast<const TdecGroups*>(groups));
  LAUNCH_CHECK();
  return SRSLTE_SUCCESS;
}
#else // !TDEC_MIX_TU: the rest of this file
} // namespace
int tdec_mix_launch(const void* args, const void* groups, unsigned waves, hipStream_t st); // tdec_mix.hip
namespace {
// int8 -> int16 widening for the 8-bit API's 16-bit fall-backs (convert_8_to_16, turbodecoder.c:451-456)
__global__ void widen_kernel(const int8_t* __restrict__ in, int16_t* __restrict__ out, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

// The same for a ragged batch: only the launched blocks of the widened groups (K <= 800 under the 8-bit API), each as far as its decoder reads -
// not every slot of the buffer (the mixed-grant line: 30.8 M elements per call, 55 us, for some 250 short blocks). Workgroup = one launched block.
__global__ __launch_bounds__(256) void widen_groups_kernel(const int8_t* __restrict__ in, int16_t* __restrict__ out, const uint32_t* __restrict__ cb_map,
                                                           uint32_t in_stride, TdecGroups gs)
{
  uint32_t j = blockIdx.x, gi = 0;
  while (gi + 1 < gs.n && j >= gs.g[gi].nof_cb) j -= gs.g[gi++].nof_cb;
  const uint32_t lcb = gs.g[gi].first_lcb + j, cb = cb_map ? cb_map[lcb] : lcb;
  uint32_t       len = (3 * (gs.g[gi].K + 32) + 12 + 31) & ~31u; // the SB layout's length covers the plain one
  len                = len < in_stride ? len : in_stride;
  const size_t o = (size_t)cb * in_stride;
  for (uint32_t i = threadIdx.x; i < len; i += 256) out[o + i] = in[o + i];
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
struct TabKey {
  uint32_t K, W, poly, nbits;
  bool     operator<(const TabKey& o) const { return memcmp(this, &o, sizeof(*this)) < 0; }
};
struct TabDev {
  uint16_t *inter, *deinter;
  uint32_t *crc_rem, *crc_rem_i;
};

} // namespace

struct srslte_hip_tdec {
  uint32_t                 max_long_cb, max_nof_cb, Kp;
  int16_t*                 d_work;
  pk_t*                    d_beta;
  int4*                    d_xy;
  pk_t*                    d_zeros;
  int16_t*                 d_conv; // widened LLRs of the 8-bit API's 16-bit fall-backs, allocated on first use
  uint32_t                 beta_stride;
  std::map<TabKey, TabDev> tabs;
  const uint32_t*          tb_rem = nullptr; // see tdec_set_tb_syndrome
  uint32_t                 tb_C   = 0;
  uint32_t*                tb_syn = nullptr;
  const uint8_t*           skip   = nullptr; // see tdec_set_skip
  const uint32_t*          cb_map = nullptr; // see tdec_set_cb_map
  uint32_t                 start_iter = 0;   // see tdec_set_resume; consumed by the next run
  uint8_t*                 tb_out = nullptr; // see tdec_set_tb_direct; consumed by the next run
  uint32_t                 tb_out_stride = 0, tb_rb = 0;
  uint8_t*                 tb_ok_out = nullptr;
  uint32_t*                d_tb_acc = nullptr;
  const uint8_t*           tb_Cof = nullptr; // tdec_set_tb_ragged; consumed by the next tdec_run_groups
  uint32_t                 tb_width = 0, tb_B = 0, tb_rows0 = 0;
  std::map<uint32_t, uint32_t*> tbA_tabs;    // K -> device CRC24A share tables of the ragged mode
  std::mutex               mtx;
};

extern "C" uint32_t srslte_hip_tdec_autoimp_get_subblocks(uint32_t K)
{ // turbodecoder.c:394-406 on an AVX2 host
  if (!(K % 16) && K > 800) return 16;
  if (!(K % 8) && K > 400) return 8;
  return 0;
}

extern "C" uint32_t srslte_hip_tdec_autoimp_get_subblocks_8bit(uint32_t K)
{ // turbodecoder.c:421-436 on an AVX2 host
  if (!(K % 32) && K > 2048) return 32;
  if (!(K % 16) && K > 800) return 16;
  if (!(K % 8) && K > 400) return 8;
  return 0;
}

extern "C" uint32_t srslte_hip_tdec_input_len(uint32_t K, int sb_layout) { return sb_layout ? 3 * (K + 32) + 12 : 3 * K + 12; }

extern "C" srslte_hip_tdec_t* srslte_hip_tdec_create(uint32_t max_long_cb, uint32_t max_nof_cb)
{
  if (max_long_cb < 40 || max_long_cb > 6144 || max_nof_cb == 0) {
    hip_log("[srslte_hip] tdec: invalid max_long_cb=%u / max_nof_cb=%u\n", max_long_cb, max_nof_cb);
    return nullptr;
  }
  auto* q         = new srslte_hip_tdec();
  q->max_long_cb  = max_long_cb;
  q->max_nof_cb   = max_nof_cb;
  q->Kp           = (max_long_cb + 16 + 31) & ~31u;
  q->beta_stride  = (max_long_cb + 8) * 64; // generic: K+4 steps; windowed: K/8+1
  q->d_work       = nullptr;
  q->d_beta       = nullptr;
  q->d_xy         = nullptr;
  q->d_conv       = nullptr;
  q->d_zeros      = nullptr;
  // windowed kernels need (K/W+1)*64 dwords per block; the generic one (K+4)*64 per 8 blocks: size for the worst
  const size_t beta_words = (size_t)max_nof_cb * (max_long_cb / 8 + 2) * 64;
  const size_t gen_words  = (size_t)((max_nof_cb + 7) / 8) * (max_long_cb + 8) * 64;
  if (hipMalloc((void**)&q->d_work, (size_t)max_nof_cb * 7 * q->Kp * sizeof(int16_t)) != hipSuccess ||
      hipMalloc((void**)&q->d_beta, sizeof(pk_t) * (beta_words > gen_words ? beta_words : gen_words)) != hipSuccess ||
      hipMalloc((void**)&q->d_xy, sizeof(int4) * (size_t)max_nof_cb * max_long_cb) != hipSuccess ||
      hipMalloc((void**)&q->d_zeros, sizeof(pk_t) * (size_t)max_long_cb) != hipSuccess ||
      hipMemset(q->d_zeros, 0, sizeof(pk_t) * (size_t)max_long_cb) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess /* the memset ran on the null stream; callers launch on non-blocking streams */) {
    hip_log("[srslte_hip] tdec: device allocation failed\n");
    if (q->d_work) (void)hipFree(q->d_work);
    delete q;
    return nullptr;
  }
  return q;
}

extern "C" void srslte_hip_tdec_destroy(srslte_hip_tdec_t* q)
{
  if (!q) return;
  for (auto& kv : q->tabs) {
    (void)hipFree(kv.second.inter);
    (void)hipFree(kv.second.deinter);
    if (kv.second.crc_rem) (void)hipFree(kv.second.crc_rem);
    if (kv.second.crc_rem_i) (void)hipFree(kv.second.crc_rem_i);
  }
  (void)hipFree(q->d_work);
  (void)hipFree(q->d_beta);
  (void)hipFree(q->d_xy);
  (void)hipFree(q->d_zeros);
  if (q->d_tb_acc) (void)hipFree(q->d_tb_acc);
  for (auto& kv : q->tbA_tabs) (void)hipFree(kv.second);
  if (q->d_conv) (void)hipFree(q->d_conv);
  delete q;
}

static int tdec_get_tables(srslte_hip_tdec_t* q, uint32_t K, uint32_t W, uint32_t poly, uint32_t nbits, TdecTables* t)
{
  std::lock_guard<std::mutex> lk(q->mtx);
  TabKey key;
  memset(&key, 0, sizeof(key));
  key.K = K; key.W = W; key.poly = poly; key.nbits = poly ? nbits : 0;
  auto it = q->tabs.find(key);
  if (it == q->tabs.end()) {
    std::vector<uint16_t> f, r;
    lte_qpp_tables(K, W, f, r);
    TabDev d = {nullptr, nullptr, nullptr, nullptr};
    HIP_TRY(hipMalloc((void**)&d.inter, K * 2));
    HIP_TRY(hipMalloc((void**)&d.deinter, K * 2));
    HIP_TRY(hipMemcpy(d.inter, f.data(), K * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d.deinter, r.data(), K * 2, hipMemcpyHostToDevice));
    if (poly) {
      // rem[j] = x^(nbits-1-j) mod g (24-bit CRC, crc.c:33-47 polynomial convention with the x^24 term included)
      std::vector<uint32_t> nat(K, 0), pos(K, 0);
      uint32_t              v = 1;
      for (int j = (int)nbits - 1; j >= 0; j--) {
        nat[j] = v;
        v <<= 1;
        if (v & 0x1000000) v ^= poly;
      }
      for (uint32_t n = 0; n < K; n++) pos[W ? (n % (K / W)) * W + n / (K / W) : n] = nat[n];
      HIP_TRY(hipMalloc((void**)&d.crc_rem, K * 4));
      HIP_TRY(hipMemcpy(d.crc_rem, pos.data(), K * 4, hipMemcpyHostToDevice));
      if (W == 16) { // sign(app1'[inter[j]]) = sign(ext2[j]): the syndrome of a DEC2 pass is taken before the interleaver (tdec_pair.inc)
        std::vector<uint32_t> posi(K);
        for (uint32_t j = 0; j < K; j++) posi[j] = pos[f[j]];
        HIP_TRY(hipMalloc((void**)&d.crc_rem_i, K * 4));
        HIP_TRY(hipMemcpy(d.crc_rem_i, posi.data(), K * 4, hipMemcpyHostToDevice));
      }
    }
    it = q->tabs.emplace(key, d).first;
  }
  t->inter   = it->second.inter;
  t->deinter = it->second.deinter;
  t->crc_rem = it->second.crc_rem;
  t->crc_rem_i = it->second.crc_rem_i;
  return SRSLTE_SUCCESS;
}

void tdec_set_skip(srslte_hip_tdec_t* q, const uint8_t* d_skip) { q->skip = d_skip; }
void tdec_set_cb_map(srslte_hip_tdec_t* q, const uint32_t* d_map) { q->cb_map = d_map; }
void tdec_set_resume(srslte_hip_tdec_t* q, uint32_t start_iter) { q->start_iter = start_iter; }

int tdec_set_tb_direct(srslte_hip_tdec_t* q, uint8_t* d_tb, uint32_t tb_stride, uint32_t payload_bytes_per_block, uint8_t* d_tb_ok)
{ // the NEXT run (16 windows, 16 bit, with tdec_set_tb_syndrome, without skip flags or a block map) assembles the transport blocks itself
  if (d_tb && !q->d_tb_acc) {
    HIP_TRY(hipMalloc((void**)&q->d_tb_acc, sizeof(uint32_t) * 4 * q->max_nof_cb));
    HIP_TRY(hipMemset(q->d_tb_acc, 0, sizeof(uint32_t) * 4 * q->max_nof_cb));
    HIP_TRY(hipDeviceSynchronize());
  }
  q->tb_out = d_tb; q->tb_out_stride = tb_stride; q->tb_rb = payload_bytes_per_block; q->tb_ok_out = d_tb_ok;
  return SRSLTE_SUCCESS;
}

// The same for a RAGGED batch (the next tdec_run_groups, 16-bit LLRs; a skipped block contributes its stored bytes): block slot cb belongs to transport-block slot cb / width,
// which has d_Cof[cb / width] code blocks; row of d_tb / d_tb_ok: the slot itself below B, rows0 + slot - B from there on.
int tdec_set_tb_ragged(srslte_hip_tdec_t* q, uint8_t* d_tb, uint32_t tb_stride, uint8_t* d_tb_ok, const uint8_t* d_Cof, uint32_t width, uint32_t B,
                       uint32_t rows0)
{
  if (int r = tdec_set_tb_direct(q, d_tb, tb_stride, 0, d_tb_ok)) return r;
  q->tb_Cof = d_Cof; q->tb_width = width; q->tb_B = B; q->tb_rows0 = rows0;
  return SRSLTE_SUCCESS;
}

// CRC24A share tables of block length K for the ragged mode, in the 16-window decoder's array order: [0, K) x^(K-1-n) mod g for every bit n of the
// block (a transport block of one block); [K, 2K) x^(K-25-n) for the payload bits n < K - 24, 0 on the block's own CRC (one of several blocks);
// [2K, 2K+16) x^(j (K-24)) mod g: block r of C contributes its payload's remainder times x^((C-1-r)(K-24))
static int tdec_tbA_table(srslte_hip_tdec_t* q, uint32_t K, const uint32_t** d_tab)
{
  auto it = q->tbA_tabs.find(K);
  if (it == q->tbA_tabs.end()) {
    std::vector<uint32_t> t(2 * (size_t)K + 16, 0), pw(K);
    uint32_t              v = 1;
    for (uint32_t i = 0; i < K; i++) { // pw[i] = x^i mod g
      pw[i] = v;
      v <<= 1;
      if (v & 0x1000000u) v ^= 0x1864CFBu;
    }
    const uint32_t Lw = K / 16;
    for (uint32_t n = 0; n < K; n++) {
      const uint32_t pos = (n % Lw) * 16 + n / Lw;
      t[pos] = pw[K - 1 - n];
      if (n + 24 < K) t[K + pos] = pw[K - 25 - n];
    }
    auto mul = [](uint32_t a, uint32_t b) {
      uint32_t r = 0;
      for (int i = 23; i >= 0; i--) {
        r <<= 1;
        if (r & 0x1000000u) r ^= 0x1864CFBu;
        if ((b >> i) & 1) r ^= a;
      }
      return r;
    };
    uint32_t step = 1; // x^(K-24) mod g = x * pw[K-25]
    step = mul(pw[K - 25], 2u);
    uint32_t f = 1;
    for (uint32_t j = 0; j < 16; j++) {
      t[2 * (size_t)K + j] = f;
      f = mul(f, step);
    }
    uint32_t* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, sizeof(uint32_t) * t.size()));
    HIP_TRY(hipMemcpy(d, t.data(), sizeof(uint32_t) * t.size(), hipMemcpyHostToDevice));
    it = q->tbA_tabs.emplace(K, d).first;
  }
  *d_tab = it->second;
  return SRSLTE_SUCCESS;
}

void tdec_set_tb_syndrome(srslte_hip_tdec_t* q, const uint32_t* d_rem, uint32_t C, uint32_t* d_syn)
{ // windowed decoders only; the caller (pdsch.hip) builds d_rem in the decoder's array order
  q->tb_rem = d_rem;
  q->tb_C   = C;
  q->tb_syn = d_syn;
}

int tdec_run_batch_w(srslte_hip_tdec_t* q, const void* d_input_any, int llr8, uint32_t in_stride, int sb_layout, uint32_t K, int force_w,
                     uint32_t nof_cb, uint32_t nof_iterations, uint32_t crc_poly, uint32_t crc_nbits, uint8_t* d_output,
                     uint32_t out_stride, uint32_t* d_iters, uint8_t* d_crc_ok, hipStream_t st)
{
  const int16_t* d_input = (const int16_t*)d_input_any;
  if (!q || !d_input || !d_output) return SRSLTE_ERROR_INVALID_INPUTS;
  if (K > q->max_long_cb) {
    hip_log("[srslte_hip] TDEC was initialized for max_long_cb=%u\n", q->max_long_cb); // turbodecoder.c:524-527
    return SRSLTE_ERROR;
  }
  const int idx = lte_cb_index(K);
  if (idx < 0 || lte_qpp_table[idx].K != K) {
    hip_log("[srslte_hip] Invalid CB length %u\n", K); // turbodecoder.c:531-534
    return SRSLTE_ERROR;
  }
  if (nof_cb > q->max_nof_cb || nof_iterations == 0 || out_stride < K / 8 || in_stride < srslte_hip_tdec_input_len(K, sb_layout) ||
      (crc_poly && (crc_nbits > K || (crc_poly >> 24) != 1)))
    return SRSLTE_ERROR_INVALID_INPUTS;
  if (nof_cb == 0) return SRSLTE_SUCCESS;
  // force_w 3016: the 16-window numerics in the state-per-lane mapping of rounds 1-2 (tdec_win_kernel<16, 0>: the tests' second opinion on
  // tdec_pair_kernel, which AUTO and 16 select)
  const bool old_map = force_w == 3016;
  if (old_map) force_w = 16;
  const uint32_t W = force_w >= 0 ? (uint32_t)force_w : (llr8 ? srslte_hip_tdec_autoimp_get_subblocks_8bit(K) : srslte_hip_tdec_autoimp_get_subblocks(K));
  if ((W != 0 && W != 8 && W != 16 && !(llr8 && W == 32)) || (W && (K % W || K / W < WIN_OVERLAP)) || (sb_layout && !W))
    return SRSLTE_ERROR_INVALID_INPUTS;
  const bool ar8 = llr8 && W >= 16; // sse8 / avx8 numerics; below that the 8-bit API widens and runs a 16-bit back-end (turbodecoder.c:465-469)
  if (llr8 && !ar8) {
    // upstream widens 3K+12 elements whatever the layout (turbodecoder.c:466), which leaves part of an SB-layout buffer stale;
    // the whole buffer is widened here
    // with a block map (ragged batches) the launched blocks sit in arbitrary slots of the buffer: every slot is widened
    const size_t per_max = ((size_t)3 * (q->max_long_cb + 32) + 12 + 31) & ~(size_t)31; // the pipelines round their strides up to 32
    const size_t per = in_stride, n = (size_t)(q->cb_map ? q->max_nof_cb : nof_cb) * per;
    if (!q->d_conv) HIP_TRY(hipMalloc((void**)&q->d_conv, (size_t)q->max_nof_cb * per_max * sizeof(int16_t)));
    if (per > per_max) return SRSLTE_ERROR_INVALID_INPUTS;
    hipLaunchKernelGGL(widen_kernel, dim3((unsigned)((n + 1023) / 1024 < 4096 ? (n + 1023) / 1024 : 4096)), dim3(256), 0, st, (const int8_t*)d_input_any,
                       q->d_conv, n);
    LAUNCH_CHECK();
    d_input = q->d_conv;
  }
  TdecArgs a;
  a.in = d_input; a.in_stride = in_stride; a.sb_layout = sb_layout; a.K = K; a.nof_cb = nof_cb; a.nof_iter = nof_iterations;
  a.work = q->d_work; a.Kp = q->Kp; a.beta = q->d_beta; a.xy = q->d_xy; a.zeros = q->d_zeros;
  a.dbg = 0;
#ifdef TDEC_DEBUG // timing experiments only (scripts/): a stray environment variable must not be able to change what a product build decodes
  a.dbg = getenv("SRSLTE_HIP_TDEC_DBG") ? atoi(getenv("SRSLTE_HIP_TDEC_DBG")) : 0;
#endif
  a.out = d_output; a.out_stride = out_stride; a.iters = d_iters; a.crc_ok = d_crc_ok;
  a.tb_rem = W ? q->tb_rem : nullptr; a.tb_C = q->tb_C ? q->tb_C : 1; a.tb_syn = q->tb_syn;
  a.skip = q->skip;
  a.cb_map = q->cb_map;
  a.start_iter = q->start_iter < nof_iterations ? q->start_iter : 0;
  q->start_iter = 0;
  a.tb_out = nullptr; a.tb_out_stride = 0; a.tb_rb = 0; a.tb_ok_out = nullptr; a.tb_acc = q->d_tb_acc;
  if (q->tb_out) { // consumed by this run, whichever kernel it takes
    const bool direct = llr8 ? ar8 : (W == 16 && !old_map); // the kernels whose last phase assembles: pair, avx8 (tdec_win_body), sse8
    if (!direct || !a.tb_rem || a.skip || a.cb_map || nof_cb % a.tb_C || a.start_iter) return SRSLTE_ERROR_INVALID_INPUTS;
    a.tb_out = q->tb_out; a.tb_out_stride = q->tb_out_stride; a.tb_rb = q->tb_rb; a.tb_ok_out = q->tb_ok_out;
    q->tb_out = nullptr;
  }
  a.prof = nullptr;
#ifdef TDEC_PROF
  static unsigned long long* d_prof = nullptr;
  if (!d_prof) HIP_TRY(hipMalloc((void**)&d_prof, sizeof(unsigned long long) * 10 * 65536));
  if (nof_cb <= 65536) a.prof = d_prof;
#endif
  int r = tdec_get_tables(q, K, W, crc_poly, crc_nbits, &a.t);
  if (r) return r;
  static const TdecGroups no_groups = {};
  if (ar8 && W == 32) {
    hipLaunchKernelGGL(tdec_ar32_kernel, dim3(nof_cb), dim3(64), 0, st, a, no_groups);
  } else if (ar8 && !old_map && TDEC_AR16_PAIR) {
    hipLaunchKernelGGL(tdec_ar16_kernel, dim3((nof_cb + 1) / 2), dim3(64), 0, st, a, no_groups);
  } else if (ar8) {
    hipLaunchKernelGGL((tdec_win_kernel<16, 1>), dim3(nof_cb), dim3(64), 0, st, a, no_groups);
  } else if (W == 16 && !old_map) {
    a.beta_stride = (K / 16 / PB + 2) * 128; // checkpoint rows of a wavefront: 64 lanes x 2 dwords
    hipLaunchKernelGGL(tdec_pair_kernel, dim3((nof_cb + 1) / 2), dim3(64), 0, st, a, no_groups);
  } else if (W == 16) {
    a.beta_stride = (K / 16 + 1) * 64;
    hipLaunchKernelGGL((tdec_win_kernel<16, 0>), dim3(nof_cb), dim3(64), 0, st, a, no_groups);
  } else if (W == 8) {
    a.beta_stride = (K / 8 + 1) * 64;
    hipLaunchKernelGGL((tdec_win_kernel<8, 0>), dim3(nof_cb), dim3(64), 0, st, a, no_groups);
  } else {
    a.beta_stride = (K + 4) * 64;
    hipLaunchKernelGGL(tdec_gen_kernel, dim3((nof_cb + 7) / 8), dim3(64), 0, st, a, no_groups);
  }
  LAUNCH_CHECK();
#ifdef TDEC_PROF
  if (a.prof && getenv("SRSLTE_HIP_TDEC_PROF")) {
    const bool pair = W == 16 && !ar8 && !old_map;
    const uint32_t nrec = pair ? (nof_cb + 1) / 2 : nof_cb; // the pair kernel writes one record per wavefront
    std::vector<unsigned long long> h((size_t)nrec * 10);
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(h.data(), a.prof, h.size() * 8, hipMemcpyDeviceToHost));
    double tot[10] = {0}, all = 0;
    for (uint32_t c = 0; c < nrec; c++) {
      for (int i = 0; i < 10; i++) tot[i] += (double)h[(size_t)c * 10 + i];
    }
    for (int i = 0; i < 10; i++) all += tot[i];
    static const char* nm_win[10] = {"extract", "elementwise", "combine", "beta warm-up", "tail+shift+beta main", "alpha warm-up", "alpha main", "crc", "decision+tb", "-"};
    static const char* nm_pair[10] = {"extract", "before sweeps", "alpha main", "after sweeps", "decisions", "beta warm-up", "tail+shift+beta main", "alpha warm-up", "-", "-"};
    const char** nm = pair ? nm_pair : nm_win;
    fprintf(stderr, "[tdec prof] K=%u W=%u blocks=%u mean cycles/block %.0f:", K, W, nrec, all / nrec);
    for (int i = 0; i < 9; i++) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * tot[i] / all);
    fprintf(stderr, "\n");
  }
#endif
  return SRSLTE_SUCCESS;
}

// One call for a ragged batch: groups[i] = nof_cb blocks of length K, in the order of the block map (tdec_set_cb_map: group i's slots behind those
// of groups 0 .. i-1). ONE launch per decoder kernel the lengths need - the two-blocks-per-wavefront kernel for every K > 800, the 8-window one,
// the unwindowed one (8-bit LLRs: the 32- and 16-window 8-bit kernels, the rest widened) - instead of one per length. Skip flags as set.
int tdec_run_groups(srslte_hip_tdec_t* q, const void* d_input_any, int llr8, uint32_t in_stride, const srslte_hip_tdec_group_t* groups, uint32_t nof_groups,
                    uint32_t nof_iterations, uint8_t* d_output, uint32_t out_stride, uint32_t* d_iters, uint8_t* d_crc_ok, hipStream_t st)
{
  if (!q || !d_input_any || !d_output || !groups || nof_groups > (uint32_t)TDEC_MAX_GROUPS || nof_iterations == 0) return SRSLTE_ERROR_INVALID_INPUTS;
  // tdec_set_tb_ragged: the decoders assemble the transport blocks (one mixed launch, 16-bit kinds only); consumed by this call whatever its outcome
  uint8_t* const       rg_tb = q->tb_out;
  const uint8_t* const rg_Cof = q->tb_Cof;
  q->tb_out = nullptr;
  q->tb_Cof = nullptr;
  const bool ragged = rg_tb != nullptr;
  if ((ragged && (!rg_Cof || llr8)) || q->start_iter || q->tb_rem) return SRSLTE_ERROR_INVALID_INPUTS; // modes of the one-length call
  enum { T_PAIR, T_WIN8, T_GEN, T_AR32, T_AR16, T_N };
  struct Plan { TdecGroups gs; uint32_t waves, stride, blocks; bool widened; };
  static const TdecGroups none = {};
  Plan     plan[T_N];
  for (auto& p_ : plan) { p_.gs = none; p_.waves = p_.stride = p_.blocks = 0; p_.widened = false; }
  uint32_t total = 0, kinds = 0; // kinds: bit per 16-bit kind present
  bool     any_widen = false;
  for (uint32_t i = 0; i < nof_groups; i++) {
    const srslte_hip_tdec_group_t& g = groups[i];
    const int idx = lte_cb_index(g.K);
    if (g.K > q->max_long_cb || idx < 0 || lte_qpp_table[idx].K != g.K) {
      hip_log("[srslte_hip] Invalid CB length %u\n", g.K);
      return SRSLTE_ERROR;
    }
    if (g.nof_cb == 0) continue;
    const uint32_t W    = llr8 ? srslte_hip_tdec_autoimp_get_subblocks_8bit(g.K) : srslte_hip_tdec_autoimp_get_subblocks(g.K);
    const bool     ar8  = llr8 && W >= 16;
    const int      type = ar8 ? (W == 32 ? T_AR32 : T_AR16) : (W == 16 ? T_PAIR : (W == 8 ? T_WIN8 : T_GEN));
    if (out_stride < g.K / 8 || in_stride < srslte_hip_tdec_input_len(g.K, W != 0) || (g.crc_poly && (g.crc_nbits > g.K || (g.crc_poly >> 24) != 1)) ||
        (type == T_GEN && g.K + 4 > q->max_long_cb / 8 + 2))
      return SRSLTE_ERROR_INVALID_INPUTS;
    // the 16-bit kinds (pair, 8-window, unwindowed - for 8-bit LLRs the widened fall-backs) go into ONE plan, slot T_PAIR, each group with its kind
    const bool mixable = type == T_PAIR || type == T_WIN8 || type == T_GEN;
    Plan&      p_ = plan[mixable ? T_PAIR : type];
    p_.gs.kind[p_.gs.n] = (uint8_t)(type == T_WIN8 ? TDEC_KIND_WIN8 : (type == T_GEN ? TDEC_KIND_GEN : TDEC_KIND_PAIR));
    p_.gs.tbA[p_.gs.n]  = nullptr;
    if (ragged && type == T_PAIR) {
      if (int r = tdec_tbA_table(q, g.K, &p_.gs.tbA[p_.gs.n])) return r;
    }
    TdecGroup& d  = p_.gs.g[p_.gs.n++];
    d.K = g.K; d.nof_cb = g.nof_cb; d.first_lcb = total; d.first_wave = p_.waves;
    if (int r = tdec_get_tables(q, g.K, W, g.crc_poly, g.crc_nbits, &d.t)) return r;
    const uint32_t waves = (type == T_PAIR || (type == T_AR16 && TDEC_AR16_PAIR)) ? (g.nof_cb + 1) / 2 : (type == T_GEN ? (g.nof_cb + 7) / 8 : g.nof_cb);
    const uint32_t stride = type == T_PAIR ? (g.K / 16 / PB + 2) * 128 : (type == T_GEN ? (g.K + 4) * 64 : (g.K / (W == 32 ? 16 : W) + 1) * 64);
    p_.waves += waves;
    p_.stride = stride > p_.stride ? stride : p_.stride;
    p_.blocks += g.nof_cb;
    p_.widened = llr8 && !ar8;
    if (mixable) kinds |= 1u << type;
    any_widen = any_widen || p_.widened;
    total += g.nof_cb;
  }
  if (total > q->max_nof_cb) return SRSLTE_ERROR_INVALID_INPUTS;
  if (total == 0) return SRSLTE_SUCCESS;
  if (any_widen) { // the 8-bit API's 16-bit fall-backs (turbodecoder.c:465-469): the blocks of the widened groups, each in its slot of the buffer
    const size_t per_max = ((size_t)3 * (q->max_long_cb + 32) + 12 + 31) & ~(size_t)31;
    if (in_stride > per_max) return SRSLTE_ERROR_INVALID_INPUTS;
    if (!q->d_conv) HIP_TRY(hipMalloc((void**)&q->d_conv, (size_t)q->max_nof_cb * per_max * sizeof(int16_t)));
    const Plan& pw = plan[T_PAIR]; // all widened groups are in the mixable plan
    hipLaunchKernelGGL(widen_groups_kernel, dim3(pw.blocks), dim3(256), 0, st, (const int8_t*)d_input_any, q->d_conv, q->cb_map, in_stride, pw.gs);
    LAUNCH_CHECK();
  }
  for (int type = 0; type < T_N; type++) {
    Plan& p_ = plan[type];
    if (!p_.gs.n) continue;
    p_.gs.xy_stride = q->max_long_cb;
    // slot T_PAIR: one kind -> that kind's own kernel, several -> the mixed launch (tdec_mix.hip)
    const int run = type != T_PAIR ? type : (ragged ? T_N : (kinds == (1u << T_PAIR) ? T_PAIR : (kinds == (1u << T_WIN8) ? T_WIN8 : (kinds == (1u << T_GEN) ? T_GEN : T_N))));
    TdecArgs a;
    memset(&a, 0, sizeof(a));
    a.in = p_.widened ? q->d_conv : (const int16_t*)d_input_any; a.in_stride = in_stride; a.sb_layout = run != T_GEN; a.nof_iter = nof_iterations;
    a.K = p_.gs.g[0].K; a.nof_cb = p_.gs.g[0].nof_cb; a.t = p_.gs.g[0].t; // replaced per wavefront (tdec_enter_group)
    a.work = q->d_work; a.Kp = q->Kp; a.beta = q->d_beta; a.xy = q->d_xy; a.zeros = q->d_zeros; a.beta_stride = p_.stride;
    a.out = d_output; a.out_stride = out_stride; a.iters = d_iters; a.crc_ok = d_crc_ok; a.tb_C = 1;
    a.skip = q->skip; a.cb_map = q->cb_map;
    if (ragged && type == T_PAIR) { // consumed below
      a.tb_out = rg_tb; a.tb_out_stride = q->tb_out_stride; a.tb_ok_out = q->tb_ok_out; a.tb_acc = q->d_tb_acc;
      a.tb_Cof = rg_Cof; a.tb_C = q->tb_width; a.tb_B = q->tb_B; a.tb_rows0 = q->tb_rows0;
    }
    switch (run) {
      case T_PAIR: hipLaunchKernelGGL(tdec_pair_kernel, dim3(p_.waves), dim3(64), 0, st, a, p_.gs); break;
      case T_WIN8: hipLaunchKernelGGL((tdec_win_kernel<8, 0>), dim3(p_.waves), dim3(64), 0, st, a, p_.gs); break;
      case T_GEN: hipLaunchKernelGGL(tdec_gen_kernel, dim3(p_.waves), dim3(64), 0, st, a, p_.gs); break;
      case T_AR32: hipLaunchKernelGGL(tdec_ar32_kernel, dim3(p_.waves), dim3(64), 0, st, a, p_.gs); break;
      case T_AR16:
        if (TDEC_AR16_PAIR) hipLaunchKernelGGL(tdec_ar16_kernel, dim3(p_.waves), dim3(64), 0, st, a, p_.gs);
        else hipLaunchKernelGGL((tdec_win_kernel<16, 1>), dim3(p_.waves), dim3(64), 0, st, a, p_.gs);
        break;
      default: // the mixed launch sets sb_layout per kind itself
        if (int r = tdec_mix_launch(&a, &p_.gs, p_.waves, st)) return r;
        break;
    }
    LAUNCH_CHECK();
  }
  return SRSLTE_SUCCESS;
}

extern "C" int srslte_hip_tdec_run_batch(srslte_hip_tdec_t* q, const int16_t* d_input, uint32_t in_stride, int sb_layout, uint32_t long_cb,
                                         uint32_t nof_cb, uint32_t nof_iterations, uint32_t crc_poly, uint32_t crc_nbits,
                                         uint8_t* d_output, uint32_t out_stride, uint32_t* d_iters, uint8_t* d_crc_ok, void* stream)
{
  return tdec_run_batch_w(q, d_input, 0, in_stride, sb_layout, long_cb, -1, nof_cb, nof_iterations, crc_poly, crc_nbits, d_output, out_stride,
                          d_iters, d_crc_ok, (hipStream_t)stream);
}

// 8-bit LLRs (srslte_tdec_run_all_8bit / srslte_tdec_iteration_8bit, turbodecoder.c:565-593): back-end per K as on an AVX2 host
extern "C" int srslte_hip_tdec_run_batch_8bit(srslte_hip_tdec_t* q, const int8_t* d_input, uint32_t in_stride, int sb_layout, uint32_t long_cb,
                                              uint32_t nof_cb, uint32_t nof_iterations, uint32_t crc_poly, uint32_t crc_nbits,
                                              uint8_t* d_output, uint32_t out_stride, uint32_t* d_iters, uint8_t* d_crc_ok, void* stream)
{
  return tdec_run_batch_w(q, d_input, 1, in_stride, sb_layout, long_cb, -1, nof_cb, nof_iterations, crc_poly, crc_nbits, d_output, out_stride,
                          d_iters, d_crc_ok, (hipStream_t)stream);
}

// srslte_tdec_init_manual equivalent (turbodecoder.c:168-215): force the numerics, W = 0 generic / 8 sse16 / 16 avx16
extern "C" int srslte_hip_tdec_run_batch_manual(srslte_hip_tdec_t* q, const int16_t* d_input, uint32_t in_stride, int sb_layout,
                                                uint32_t long_cb, uint32_t nof_subblocks, uint32_t nof_cb, uint32_t nof_iterations,
                                                uint32_t crc_poly, uint32_t crc_nbits, uint8_t* d_output, uint32_t out_stride,
                                                uint32_t* d_iters, uint8_t* d_crc_ok, void* stream)
{
  return tdec_run_batch_w(q, d_input, 0, in_stride, sb_layout, long_cb, (int)nof_subblocks, nof_cb, nof_iterations, crc_poly, crc_nbits,
                          d_output, out_stride, d_iters, d_crc_ok, (hipStream_t)stream);
}
#endif // TDEC_MIX_TU
