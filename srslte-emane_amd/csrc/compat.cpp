// The reference's single-call C API (include/srslte_hip/srslte_compat.h) as synchronous host wrappers over the HIP
// kernels: copy in -> launch on the default stream -> copy out. One object per host thread, as upstream
// (SURVEY §8b "Threading"); staging buffers are per object, so distinct objects may be used from distinct threads.
#include "phy_hip_internal.hpp"
#include "srslte_hip/srslte_compat.h"
#include <atomic>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <vector>

#define ERROR(fmt, ...) hip_log("[srslte_hip] " fmt "\n", ##__VA_ARGS__)

namespace {

struct DevStage { // grow-only device staging buffer
  void*  ptr   = nullptr;
  size_t bytes = 0;
  void*  get(size_t n)
  {
    if (n > bytes) {
      if (ptr) (void)hipFree(ptr);
      ptr   = nullptr;
      bytes = 0;
      if (hipMalloc(&ptr, n) != hipSuccess) {
        ERROR("hipMalloc(%zu) failed", n);
        return nullptr;
      }
      bytes = n;
    }
    return ptr;
  }
  void release()
  {
    if (ptr) (void)hipFree(ptr);
    ptr   = nullptr;
    bytes = 0;
  }
};

// Host <-> device traffic of the single-call API. Every host thread owns a non-blocking stream and a pinned bounce arena: a call copies its
// operands into the arena, queues the copies, the kernels and the copies back on that stream and waits once at the end. (hipMemcpy on the
// callers' pageable buffers costs ~50 us per call and serialises every host thread on the null stream; the reference's worker threads -
// three sf_workers in srsue - call these functions concurrently on distinct objects, SURVEY 8b "Threading".)
std::atomic<unsigned long long> g_stats[4];
void stats_at_exit()
{
  fprintf(stderr, "[srslte_hip] stats: stream_waits=%llu dlsch_decode2=%llu tdec_single_block_calls=%llu\n", g_stats[0].load(), g_stats[1].load(),
          g_stats[2].load());
}
struct StatsInit {
  StatsInit()
  {
    if (getenv("SRSLTE_HIP_STATS")) atexit(stats_at_exit);
  }
} g_stats_init;

struct HostLink {
  hipStream_t st  = nullptr;
  uint8_t*    pin = nullptr;
  size_t      cap = 0, used = 0;
  struct Back { void* host; const uint8_t* pinned; size_t n; };
  std::vector<Back> back; // device -> host copies queued on the stream: their host-side half happens in flush()
  bool ok = true;
  hipStream_t stream()
  {
    if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
    return st;
  }
  uint8_t* take(size_t n)
  { // n bytes of the arena; growing it waits for what is queued (the old arena is still being read / written)
    n = (n + 255) & ~(size_t)255;
    if (used + n > cap) {
      if (!flush()) return nullptr;
      const size_t want = (cap * 2 > n ? cap * 2 : n) + (1u << 20);
      if (pin) (void)hipHostFree(pin);
      pin = nullptr;
      cap = 0;
      if (hipHostMalloc((void**)&pin, want) != hipSuccess) {
        ERROR("hipHostMalloc(%zu) failed", want);
        return nullptr;
      }
      cap = want;
    }
    uint8_t* p = pin + used;
    used += n;
    return p;
  }
  bool flush()
  { // wait for the stream, hand the results to the caller's buffers, empty the arena
    bool r = ok && (!st || hipStreamSynchronize(st) == hipSuccess);
    g_stats[0]++;
    for (const Back& b : back) {
      if (r) memcpy(b.host, b.pinned, b.n);
    }
    back.clear();
    used = 0;
    ok   = true;
    return r;
  }
  ~HostLink()
  {
    if (st) (void)hipStreamDestroy(st);
    if (pin) (void)hipHostFree(pin);
  }
};
thread_local HostLink g_link;
void* tl_stream() { return g_link.stream(); }

bool h2d(void* d, const void* h, size_t n)
{
  if (n == 0) return true;
  uint8_t* p = g_link.take(n);
  if (!p || !g_link.stream()) return false;
  memcpy(p, h, n);
  const bool r = hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, g_link.st) == hipSuccess;
  g_link.ok    = g_link.ok && r;
  return r;
}
// queues the copy back; the data is in `h` after the next flush (d2h() below does both)
bool d2h_later(void* h, const void* d, size_t n)
{
  if (n == 0) return true;
  uint8_t* p = g_link.take(n);
  if (!p || !g_link.stream()) return false;
  const bool r = hipMemcpyAsync(p, d, n, hipMemcpyDeviceToHost, g_link.st) == hipSuccess;
  g_link.ok    = g_link.ok && r;
  if (r) g_link.back.push_back({h, p, n});
  return r;
}
bool d2h(void* h, const void* d, size_t n) { return d2h_later(h, d, n) && g_link.flush(); }
// Small operands (a code block: under 1 KB in, 1.5 KB out): the kernel reads and writes the pinned arena itself - device-visible host memory -
// instead of two DMA operations queued around it (each costs more than the kernel: rocprofv3 on phy_dl_test, profiles/r04/dropin_phy_dl_test_trace.txt).
// zc_in: a device-readable copy of h[0, n); zc_out: n device-writable bytes that flush() hands to `h`.
const uint8_t* zc_in(const void* h, size_t n)
{
  uint8_t* p = g_link.take(n);
  if (!p || !g_link.stream()) return nullptr;
  memcpy(p, h, n);
  return p;
}
uint8_t* zc_out(void* h, size_t n)
{
  uint8_t* p = g_link.take(n);
  if (!p || !g_link.stream()) return nullptr;
  g_link.back.push_back({h, p, n});
  return p;
}

void* host_alloc(size_t n)
{ // srslte_vec_malloc: posix_memalign to the SIMD width (vector.c:118-125)
  void* p = nullptr;
  if (posix_memalign(&p, 64, n ? n : 64)) return nullptr;
  return p;
}

// ---- state behind srslte_dft_plan_t.p
struct DftState {
  DevStage in, out;
  cf_t *   g_in = nullptr, *g_out = nullptr; // guru: caller buffers captured at plan time (dft_fftw.c:137-165)
  int      how_many = 0, idist = 0, odist = 0, istride = 1, ostride = 1;
  std::vector<cf_t> pack_in, pack_out; // guru plans with an element stride: contiguous host images of the strided caller buffers
};

// ---- state behind srslte_ofdm_t.fft_plan.p
struct OfdmState {
  srslte_hip_ofdm_t* h = nullptr;
  DevStage           in, out;
  bool               is_rx = true, norm = false, shift = false;
  float              shift_f = 0.f;
  int                mbsfn_region = 0; // 0 = regular layout on the device object
};

struct TdecState {
  srslte_hip_tdec_t* h = nullptr;
  DevStage           in, out;
  // srslte_dlsch_decode2 on the srslte_sch_t this decoder is embedded in: the transport-block decoder, made on first use and again when a
  // larger block or the other LLR width comes
  srslte_hip_sch_t*  sch = nullptr;
  uint32_t           sch_tbs = 0, sch_e = 0;
  bool               sch_l8  = false;
  std::vector<uint8_t> cb_bytes, cb_crc;
};

struct ChestState {
  srslte_hip_chest_dl_t* h = nullptr;
  DevStage               grid, ce, res;
};

// demod/tcod calls have no object to hang device staging on: one set per host thread, so that worker threads can call them
// concurrently as they can upstream (SURVEY §8b "Threading")
thread_local DevStage g_demod_in, g_demod_out, g_tcod_in, g_tcod_out;

int cp_nsymb(srslte_cp_t cp) { return cp == SRSLTE_CP_NORM ? 7 : 6; }

} // namespace

extern "C" void* srslte_hip_compat_thread_stream(unsigned* flags)
{
  hipStream_t st = g_link.stream();
  if (flags) {
    *flags = 0;
    if (st) (void)hipStreamGetFlags(st, flags);
  }
  return st;
}
extern "C" void srslte_hip_compat_stats(unsigned long long out[4])
{
  for (int i = 0; i < 4; i++) out[i] = g_stats[i].load();
}

extern "C" {

// phy_common.c:294-345. When the reference's own common/phy_common.c is linked into the program its definitions are the ones
// in effect, for this library's calls too (ordinary symbol interposition): the OFDM objects below ask srslte_symbol_sz(), not a table of their own.
static bool g_use_standard_rates = false;
void srslte_use_standard_symbol_size(bool enabled) { g_use_standard_rates = enabled; }
int  srslte_symbol_sz_power2(uint32_t nof_prb)
{
  if (nof_prb <= 6) return 128;
  if (nof_prb <= 15) return 256;
  if (nof_prb <= 25) return 512;
  if (nof_prb <= 50) return 1024;
  if (nof_prb <= 75) return 1536;
  if (nof_prb <= 110) return 2048;
  return -1;
}
int srslte_symbol_sz(uint32_t nof_prb)
{
  if (nof_prb == 0) return SRSLTE_ERROR;
  return g_use_standard_rates ? srslte_symbol_sz_power2(nof_prb) : lte_symbol_sz((int)nof_prb);
}

// ====================================================================================================== DFT
void srslte_dft_load(void) {} // FFTW wisdom import/export (dft_fftw.c:44-57): nothing to persist
void srslte_dft_exit(void) {}

int srslte_dft_plan_c(srslte_dft_plan_t* plan, int dft_points, srslte_dft_dir_t dir)
{ // dft_fftw.c:167-191
  if (!plan) return SRSLTE_ERROR_INVALID_INPUTS;
  FftFactors  f;
  const cf32* tw;
  if (fft_get_plan(dft_points, &f, &tw)) return SRSLTE_ERROR;
  memset(plan, 0, sizeof(*plan));
  plan->in  = host_alloc(sizeof(cf_t) * dft_points);
  plan->out = host_alloc(sizeof(cf_t) * dft_points);
  plan->p   = new DftState();
  plan->size = plan->init_size = dft_points;
  plan->mode    = SRSLTE_DFT_COMPLEX;
  plan->dir     = dir;
  plan->forward = dir == SRSLTE_DFT_FORWARD;
  return SRSLTE_SUCCESS;
}

int srslte_dft_plan_r(srslte_dft_plan_t* plan, int dft_points, srslte_dft_dir_t dir)
{ // dft_fftw.c:209-232: FFTW r2r plan, R2HC (forward) / HC2R (backward); served by the complex kernel
  if (!plan) return SRSLTE_ERROR_INVALID_INPUTS;
  FftFactors  f;
  const cf32* tw;
  if (fft_get_plan(dft_points, &f, &tw)) return SRSLTE_ERROR;
  memset(plan, 0, sizeof(*plan));
  plan->in  = host_alloc(sizeof(float) * dft_points);
  plan->out = host_alloc(sizeof(float) * dft_points);
  plan->p   = new DftState();
  plan->size = plan->init_size = dft_points;
  plan->mode    = SRSLTE_REAL;
  plan->dir     = dir;
  plan->forward = dir == SRSLTE_DFT_FORWARD;
  return SRSLTE_SUCCESS;
}

int srslte_dft_plan(srslte_dft_plan_t* plan, int dft_points, srslte_dft_dir_t dir, srslte_dft_mode_t mode)
{ // dft_fftw.c:80-85
  return mode == SRSLTE_DFT_COMPLEX ? srslte_dft_plan_c(plan, dft_points, dir) : srslte_dft_plan_r(plan, dft_points, dir);
}

int srslte_dft_plan_guru_c(srslte_dft_plan_t* plan, int dft_points, srslte_dft_dir_t dir, cf_t* in_buffer, cf_t* out_buffer, int istride,
                           int ostride, int how_many, int idist, int odist)
{ // dft_fftw.c:137-165: batched strided transform bound to caller buffers
  if (!plan || !in_buffer || !out_buffer) return SRSLTE_ERROR_INVALID_INPUTS;
  if (istride < 1 || ostride < 1 || how_many < 1) return SRSLTE_ERROR_INVALID_INPUTS;
  FftFactors  f;
  const cf32* tw;
  if (fft_get_plan(dft_points, &f, &tw)) return SRSLTE_ERROR;
  memset(plan, 0, sizeof(*plan));
  auto* st     = new DftState();
  st->g_in     = in_buffer;
  st->g_out    = out_buffer;
  st->how_many = how_many;
  st->idist    = idist;
  st->odist    = odist;
  st->istride  = istride;
  st->ostride  = ostride;
  plan->p      = st;
  plan->size = plan->init_size = dft_points;
  plan->mode    = SRSLTE_DFT_COMPLEX;
  plan->dir     = dir;
  plan->forward = dir == SRSLTE_DFT_FORWARD;
  plan->is_guru = true;
  return SRSLTE_SUCCESS;
}

int srslte_dft_replan_c(srslte_dft_plan_t* plan, int new_dft_points)
{ // dft_fftw.c:120-135
  FftFactors  f;
  const cf32* tw;
  if (!plan || fft_get_plan(new_dft_points, &f, &tw)) return SRSLTE_ERROR;
  if (new_dft_points > plan->init_size) {
    free(plan->in);
    free(plan->out);
    plan->in        = host_alloc(sizeof(cf_t) * new_dft_points);
    plan->out       = host_alloc(sizeof(cf_t) * new_dft_points);
    plan->init_size = new_dft_points;
  }
  plan->size = new_dft_points;
  return SRSLTE_SUCCESS;
}

int srslte_dft_replan_r(srslte_dft_plan_t* plan, int new_dft_points)
{ // dft_fftw.c:193-207 (the host buffers keep their init-time size upstream; grown here when needed)
  FftFactors  f;
  const cf32* tw;
  if (!plan || fft_get_plan(new_dft_points, &f, &tw)) return SRSLTE_ERROR;
  if (new_dft_points > plan->init_size) {
    free(plan->in);
    free(plan->out);
    plan->in        = host_alloc(sizeof(float) * new_dft_points);
    plan->out       = host_alloc(sizeof(float) * new_dft_points);
    plan->init_size = new_dft_points;
  }
  plan->size = new_dft_points;
  return SRSLTE_SUCCESS;
}

int srslte_dft_replan(srslte_dft_plan_t* plan, int new_dft_points)
{ // dft_fftw.c:66-78
  if (new_dft_points <= plan->init_size) {
    return plan->mode == SRSLTE_DFT_COMPLEX ? srslte_dft_replan_c(plan, new_dft_points) : srslte_dft_replan_r(plan, new_dft_points);
  }
  ERROR("DFT: Error calling replan: new_dft_points (%d) must be lower or equal dft_size passed initially (%d)", new_dft_points, plan->init_size);
  return SRSLTE_ERROR;
}

int srslte_dft_replan_guru_c(srslte_dft_plan_t* plan, int new_dft_points, cf_t* in_buffer, cf_t* out_buffer, int istride, int ostride,
                             int how_many, int idist, int odist)
{ // dft_fftw.c:93-118
  if (!plan || !plan->p) return SRSLTE_ERROR_INVALID_INPUTS;
  const srslte_dft_dir_t dir = plan->dir;
  const bool             norm = plan->norm, dc = plan->dc, mirror = plan->mirror;
  delete (DftState*)plan->p;
  int r = srslte_dft_plan_guru_c(plan, new_dft_points, dir, in_buffer, out_buffer, istride, ostride, how_many, idist, odist);
  plan->norm = norm; plan->dc = dc; plan->mirror = mirror;
  return r;
}

void srslte_dft_plan_set_mirror(srslte_dft_plan_t* plan, bool val) { plan->mirror = val; }
void srslte_dft_plan_set_db(srslte_dft_plan_t* plan, bool val) { plan->db = val; }
void srslte_dft_plan_set_norm(srslte_dft_plan_t* plan, bool val) { plan->norm = val; }
void srslte_dft_plan_set_dc(srslte_dft_plan_t* plan, bool val) { plan->dc = val; }

void srslte_dft_plan_free(srslte_dft_plan_t* plan)
{ // dft_fftw.c:336-347
  if (!plan || !plan->size) return;
  if (!plan->is_guru) {
    free(plan->in);
    free(plan->out);
  }
  if (plan->p) {
    auto* st = (DftState*)plan->p;
    st->in.release();
    st->out.release();
    delete st;
  }
  memset(plan, 0, sizeof(*plan));
}

static void dft_exec(srslte_dft_plan_t* plan, const cf_t* in, cf_t* out, int howmany, int idist, int odist)
{
  auto*        st = (DftState*)plan->p;
  const size_t nin = sizeof(cf_t) * ((size_t)(howmany - 1) * idist + plan->size), nout = sizeof(cf_t) * ((size_t)(howmany - 1) * odist + plan->size);
  void *       di = st->in.get(nin), *dout = st->out.get(nout);
  if (!di || !dout || !h2d(di, in, nin)) return;
  if (srslte_hip_dft_batch(di, dout, plan->size, howmany, idist, odist, plan->forward ? 1 : 0, 1.0f, tl_stream())) return;
  if (howmany > 1 && odist != plan->size) {
    for (int i = 0; i < howmany; i++) d2h(out + (size_t)i * odist, (char*)dout + sizeof(cf_t) * (size_t)i * odist, sizeof(cf_t) * plan->size);
  } else {
    d2h(out, dout, nout);
  }
}

void srslte_dft_run_c_zerocopy(srslte_dft_plan_t* plan, const cf_t* in, cf_t* out) { dft_exec(plan, in, out, 1, plan->size, plan->size); } // dft_fftw.c:277-279

void srslte_dft_run_c(srslte_dft_plan_t* plan, const cf_t* in, cf_t* out)
{ // dft_fftw.c:281-305: copy_pre (mirror/dc for BACKWARD) -> transform -> norm -> dB -> copy_post (mirror for FORWARD)
  const int N = plan->size, offset = plan->dc ? 1 : 0;
  cf_t *    pin = (cf_t*)plan->in, *pout = (cf_t*)plan->out;
  if (plan->mirror && !plan->forward) { // dft_fftw.c:249-260
    const int hlen = N / 2;
    memset(pin, 0, sizeof(cf_t) * offset);
    memcpy(&pin[offset], &in[hlen], sizeof(cf_t) * (N - hlen - offset));
    memcpy(&pin[N - hlen], in, sizeof(cf_t) * hlen);
  } else {
    memcpy(pin, in, sizeof(cf_t) * N);
  }
  dft_exec(plan, pin, pout, 1, N, N);
  float* f = (float*)pout;
  if (plan->norm) {
    const float norm = 1.0f / sqrtf((float)N);
    for (int i = 0; i < 2 * N; i++) f[i] *= norm;
  }
  if (plan->db) { // dft_fftw.c:298-302: 10*log10 of the complex value; only the real-magnitude use survives upstream
    for (int i = 0; i < N; i++) {
      f[2 * i]     = 10.0f * log10f(hypotf(f[2 * i], f[2 * i + 1]));
      f[2 * i + 1] = 0.f;
    }
  }
  if (plan->mirror && plan->forward) { // dft_fftw.c:262-272
    const int hlen = (N + 1) / 2;
    memcpy(out, &pout[hlen], sizeof(cf_t) * (N - hlen));
    memcpy(&out[N - hlen], &pout[offset], sizeof(cf_t) * (hlen - offset));
  } else {
    memcpy(out, pout, sizeof(cf_t) * N);
  }
}

void srslte_dft_run_r(srslte_dft_plan_t* plan, const float* in, float* out)
{ // dft_fftw.c:315-334. R2HC: out = r0 r1 .. r[N/2] i[(N+1)/2-1] .. i1 of the forward DFT; HC2R: its unnormalised inverse.
  const int         N = plan->size;
  std::vector<cf_t> a(N), b(N);
  if (plan->forward) {
    for (int i = 0; i < N; i++) a[i] = {in[i], 0.f};
  } else {
    a[0] = {in[0], 0.f};
    for (int k = 1; k < N - k; k++) {
      a[k]     = {in[k], in[N - k]};
      a[N - k] = {in[k], -in[N - k]};
    }
    if (N % 2 == 0) a[N / 2] = {in[N / 2], 0.f};
  }
  dft_exec(plan, a.data(), b.data(), 1, N, N);
  float* f = (float*)plan->out;
  if (plan->forward) {
    for (int k = 0; k <= N / 2; k++) f[k] = b[k].re;
    for (int k = 1; k < N - k; k++) f[N - k] = b[k].im;
  } else {
    for (int i = 0; i < N; i++) f[i] = b[i].re;
  }
  if (plan->norm) {
    const float norm = 1.0f / N;
    for (int i = 0; i < N; i++) f[i] *= norm;
  }
  if (plan->db) {
    for (int i = 0; i < N; i++) f[i] = 10.0f * log10f(f[i]);
  }
  memcpy(out, f, sizeof(float) * N);
}

void srslte_dft_run(srslte_dft_plan_t* plan, const void* in, void* out)
{ // dft_fftw.c:269-275
  if (plan->mode == SRSLTE_DFT_COMPLEX) {
    srslte_dft_run_c(plan, (const cf_t*)in, (cf_t*)out);
  } else {
    srslte_dft_run_r(plan, (const float*)in, (float*)out);
  }
}

void srslte_dft_run_guru_c(srslte_dft_plan_t* plan)
{ // dft_fftw.c:307-313
  if (!plan->is_guru) {
    ERROR("srslte_dft_run_guru_c: the selected plan is not guru!");
    return;
  }
  auto* st = (DftState*)plan->p;
  if (st->istride == 1 && st->ostride == 1) {
    dft_exec(plan, st->g_in, st->g_out, st->how_many, st->idist, st->odist);
    return;
  }
  // fftw_plan_many_dft's general layout: element j of transform i at in[i * idist + j * istride] (and the same with odist / ostride on the
  // way out). The device kernels read unit-stride rows, so the host side packs the strided caller buffer into rows of N, transforms them as
  // one batch and scatters the rows back; only the addressed elements of the caller's output are written, as FFTW does.
  const int    N = plan->size, M = st->how_many;
  st->pack_in.resize((size_t)N * M);
  st->pack_out.resize((size_t)N * M);
  for (int i = 0; i < M; i++) {
    const cf_t* src = st->g_in + (size_t)i * st->idist;
    cf_t*       dst = st->pack_in.data() + (size_t)i * N;
    for (int j = 0; j < N; j++) dst[j] = src[(size_t)j * st->istride];
  }
  dft_exec(plan, st->pack_in.data(), st->pack_out.data(), M, N, N);
  for (int i = 0; i < M; i++) {
    const cf_t* src = st->pack_out.data() + (size_t)i * N;
    cf_t*       dst = st->g_out + (size_t)i * st->odist;
    for (int j = 0; j < N; j++) dst[(size_t)j * st->ostride] = src[j];
  }
}

// ====================================================================================================== OFDM
static int ofdm_init(srslte_ofdm_t* q, srslte_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, uint32_t nof_prb, bool rx, srslte_sf_t sf_type, int symbol_sz = 0)
{ // ofdm.c:43-133; symbol_sz 0: srslte_symbol_sz(nof_prb) as srslte_ofdm_rx_init / tx_init / set_prb ask for it (ofdm.c:235-273,:307-350)
  const int N = symbol_sz > 0 ? symbol_sz : srslte_symbol_sz(nof_prb);
  if (!q || N < 0) {
    ERROR("Error: Invalid nof_prb=%u", nof_prb);
    return SRSLTE_ERROR;
  }
  memset(q, 0, sizeof(*q));
  q->max_prb = nof_prb;
  q->symbol_sz = (uint32_t)N; q->nof_symbols = (uint32_t)cp_nsymb(cp); q->nof_symbols_mbsfn = 6; q->cp = cp;
  q->nof_re = 12 * nof_prb; q->nof_guards = (N - q->nof_re) / 2; q->slot_sz = 15 * N / 2; q->sf_sz = 15 * N;
  q->in_buffer = in_buffer; q->out_buffer = out_buffer;
  q->fft_plan.size = q->fft_plan.init_size = N;
  q->fft_plan.dir     = rx ? SRSLTE_DFT_FORWARD : SRSLTE_DFT_BACKWARD;
  q->fft_plan.forward = rx;
  q->fft_plan.mirror  = true; // ofdm.c:110-111
  q->fft_plan.dc      = true;
  auto* st            = new OfdmState();
  st->is_rx           = rx;
  st->h               = srslte_hip_ofdm_create_sz((int)nof_prb, N, cp == SRSLTE_CP_NORM, rx);
  if (!st->h) {
    delete st;
    return SRSLTE_ERROR;
  }
  q->fft_plan.p = st;
  if (in_buffer) bzero(in_buffer, sizeof(cf_t) * (rx ? q->sf_sz : 2 * q->nof_symbols * q->nof_re)); // ofdm.c:80-84
  if (sf_type == SRSLTE_SF_MBSFN) { // ofdm.c:123-130
    q->mbsfn_subframe   = true;
    q->non_mbsfn_region = 2;
  }
  return SRSLTE_SUCCESS;
}

int srslte_ofdm_init_mbsfn_(srslte_ofdm_t* q, srslte_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, int symbol_sz, int nof_prb, srslte_dft_dir_t dir,
                            srslte_sf_t sf_type)
{ // ofdm.c:43-133: the symbol size is the caller's (srslte_symbol_sz of either rate family in every upstream caller)
  if (symbol_sz <= 0 || nof_prb <= 0) {
    ERROR("Error: Invalid symbol_sz=%d / nof_prb=%d", symbol_sz, nof_prb);
    return SRSLTE_ERROR;
  }
  return ofdm_init(q, cp, in_buffer, out_buffer, (uint32_t)nof_prb, dir == SRSLTE_DFT_FORWARD, sf_type, symbol_sz);
}
int srslte_ofdm_init_(srslte_ofdm_t* q, srslte_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, int symbol_sz, int nof_prb, srslte_dft_dir_t dir)
{ // ofdm.c:38-40
  return srslte_ofdm_init_mbsfn_(q, cp, in_buffer, out_buffer, symbol_sz, nof_prb, dir, SRSLTE_SF_NORM);
}

int srslte_ofdm_rx_init(srslte_ofdm_t* q, srslte_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, uint32_t max_prb) { return ofdm_init(q, cp, in_buffer, out_buffer, max_prb, true, SRSLTE_SF_NORM); }
int srslte_ofdm_tx_init(srslte_ofdm_t* q, srslte_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, uint32_t nof_prb) { return ofdm_init(q, cp, in_buffer, out_buffer, nof_prb, false, SRSLTE_SF_NORM); }
int srslte_ofdm_rx_init_mbsfn(srslte_ofdm_t* q, srslte_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, uint32_t max_prb) { return ofdm_init(q, cp, in_buffer, out_buffer, max_prb, true, SRSLTE_SF_MBSFN); }   // ofdm.c:246-256
int srslte_ofdm_tx_init_mbsfn(srslte_ofdm_t* q, srslte_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, uint32_t nof_prb) { return ofdm_init(q, cp, in_buffer, out_buffer, nof_prb, false, SRSLTE_SF_MBSFN); } // ofdm.c:284-305
void srslte_ofdm_set_non_mbsfn_region(srslte_ofdm_t* q, uint8_t non_mbsfn_region) { q->non_mbsfn_region = non_mbsfn_region; }                                                                      // ofdm.c:133-136

static void ofdm_free(srslte_ofdm_t* q)
{ // ofdm.c:214-233
  if (!q) return;
  auto* st = (OfdmState*)q->fft_plan.p;
  if (st) {
    srslte_hip_ofdm_destroy(st->h);
    st->in.release();
    st->out.release();
    delete st;
  }
  memset(q, 0, sizeof(*q));
}
void srslte_ofdm_rx_free(srslte_ofdm_t* q) { ofdm_free(q); }
void srslte_ofdm_tx_free(srslte_ofdm_t* q) { ofdm_free(q); }

static int ofdm_set_prb(srslte_ofdm_t* q, srslte_cp_t cp, uint32_t nof_prb, bool rx)
{ // ofdm.c:307-350
  if (nof_prb > q->max_prb) {
    ERROR("OFDM: Error calling set_prb: nof_prb (%u) must be equal or lower initialized max_prb (%u)", nof_prb, q->max_prb);
    return SRSLTE_ERROR;
  }
  const uint32_t max_prb = q->max_prb;
  cf_t *         in = q->in_buffer, *out = q->out_buffer;
  const bool     norm = q->fft_plan.norm, shift = q->freq_shift, mbsfn = q->mbsfn_subframe;
  const float    sf = q->freq_shift_f;
  const uint8_t  region = q->non_mbsfn_region;
  ofdm_free(q);
  if (ofdm_init(q, cp, in, out, nof_prb, rx, mbsfn ? SRSLTE_SF_MBSFN : SRSLTE_SF_NORM)) return SRSLTE_ERROR;
  q->max_prb          = max_prb;
  q->fft_plan.norm    = norm;
  q->non_mbsfn_region = region;
  if (shift) srslte_ofdm_set_freq_shift(q, sf);
  return SRSLTE_SUCCESS;
}
int srslte_ofdm_rx_set_prb(srslte_ofdm_t* q, srslte_cp_t cp, uint32_t nof_prb) { return ofdm_set_prb(q, cp, nof_prb, true); }
int srslte_ofdm_tx_set_prb(srslte_ofdm_t* q, srslte_cp_t cp, uint32_t nof_prb) { return ofdm_set_prb(q, cp, nof_prb, false); }

int srslte_ofdm_set_freq_shift(srslte_ofdm_t* q, float freq_shift)
{ // ofdm.c:360-378
  q->fft_plan.dc  = false;
  q->freq_shift   = true;
  q->freq_shift_f = freq_shift;
  return SRSLTE_SUCCESS;
}
void srslte_ofdm_set_normalize(srslte_ofdm_t* q, bool normalize_enable) { q->fft_plan.norm = normalize_enable; } // ofdm.c:576-578

static bool ofdm_sync_state(srslte_ofdm_t* q, OfdmState* st)
{
  if (st->norm != q->fft_plan.norm) {
    st->norm = q->fft_plan.norm;
    if (srslte_hip_ofdm_set_normalize(st->h, st->norm)) return false;
  }
  if (q->freq_shift && (!st->shift || st->shift_f != q->freq_shift_f)) {
    st->shift   = true;
    st->shift_f = q->freq_shift_f;
    if (srslte_hip_ofdm_set_freq_shift(st->h, st->shift_f)) return false;
  }
  const int region = q->mbsfn_subframe ? q->non_mbsfn_region : 0;
  if (region != st->mbsfn_region) {
    if (srslte_hip_ofdm_set_mbsfn(st->h, region != 0, region)) return false;
    st->mbsfn_region = region;
  }
  return true;
}

// Guard between the non-MBSFN and the MBSFN region of slot 0 (phy_common.h:147): never written by the transmitter (ofdm.c:570-572)
static void mbsfn_gap(const srslte_ofdm_t* q, size_t* begin, size_t* len)
{
  const int N = (int)q->symbol_sz, ext = lte_cp_len_ext(N), n0 = lte_cp_len_norm(0, N), n1 = lte_cp_len_norm(1, N);
  if (q->non_mbsfn_region == 1) {
    *begin = (size_t)(n0 + N);
    *len   = (size_t)(ext - n0);
  } else {
    *begin = (size_t)(n0 + N + n1 + N);
    *len   = (size_t)(2 * ext - n0 - n1);
  }
}

// One launch over `slots` (0 = slot 0, 1 = slot 1, 2 = both). host_in/host_out point at the first processed slot;
// dev_slot is where that slot sits in the device staging (the geometry is relative to the subframe base).
static void ofdm_run(srslte_ofdm_t* q, const cf_t* host_in, cf_t* host_out, int slots, int dev_slot, bool mbsfn_layout)
{
  auto* st = (OfdmState*)q->fft_plan.p;
  if (!st || !ofdm_sync_state(q, st)) return;
  const size_t t_slot = sizeof(cf_t) * q->slot_sz, g_slot = sizeof(cf_t) * q->nof_symbols * q->nof_re;
  const size_t in_slot = st->is_rx ? t_slot : g_slot, out_slot = st->is_rx ? g_slot : t_slot;
  const int    nslots = slots == 2 ? 2 : 1;
  char *       di = (char*)st->in.get(2 * in_slot), *dout = (char*)st->out.get(2 * out_slot);
  if (!di || !dout || !h2d(di + dev_slot * in_slot, host_in, nslots * in_slot)) return;
  int r;
  if (slots == 2) {
    r = st->is_rx ? srslte_hip_ofdm_rx_sf_batch(st->h, di, dout, 1, tl_stream()) : srslte_hip_ofdm_tx_sf_batch(st->h, di, dout, 1, tl_stream());
  } else {
    r = srslte_hip_ofdm_slot_batch(st->h, di, dout, 1, dev_slot, mbsfn_layout, tl_stream());
  }
  if (r) return;
  if (!st->is_rx && mbsfn_layout && dev_slot == 0) { // leave the caller's guard samples untouched, as upstream
    size_t gb, gl;
    mbsfn_gap(q, &gb, &gl);
    d2h(host_out, dout, sizeof(cf_t) * gb);
    d2h(host_out + gb + gl, dout + sizeof(cf_t) * (gb + gl), nslots * out_slot - sizeof(cf_t) * (gb + gl));
  } else {
    d2h(host_out, dout + dev_slot * out_slot, nslots * out_slot);
  }
}

void srslte_ofdm_rx_sf(srslte_ofdm_t* q) { ofdm_run(q, q->in_buffer, q->out_buffer, 2, 0, q->mbsfn_subframe); } // ofdm.c:453-467
void srslte_ofdm_tx_sf(srslte_ofdm_t* q) { ofdm_run(q, q->in_buffer, q->out_buffer, 2, 0, q->mbsfn_subframe); } // ofdm.c:580-594
void srslte_ofdm_rx_sf_ng(srslte_ofdm_t* q, cf_t* input, cf_t* output)
{ // ofdm.c:469-483: the MBSFN branch upstream ignores the arguments and works on the bound buffers
  if (q->mbsfn_subframe) {
    ofdm_run(q, q->in_buffer, q->out_buffer, 2, 0, true);
  } else {
    ofdm_run(q, input, output, 2, 0, false);
  }
}
void srslte_ofdm_rx_slot(srslte_ofdm_t* q, int slot_in_sf)
{ // ofdm.c:398-422
  ofdm_run(q, q->in_buffer + slot_in_sf * q->slot_sz, q->out_buffer + slot_in_sf * q->nof_re * q->nof_symbols, slot_in_sf ? 1 : 0, slot_in_sf ? 1 : 0, false);
}
void srslte_ofdm_tx_slot(srslte_ofdm_t* q, int slot_in_sf)
{ // ofdm.c:488-530
  ofdm_run(q, q->in_buffer + slot_in_sf * q->nof_re * q->nof_symbols, q->out_buffer + slot_in_sf * q->slot_sz, slot_in_sf ? 1 : 0, slot_in_sf ? 1 : 0, false);
}
void srslte_ofdm_rx_slot_ng(srslte_ofdm_t* q, cf_t* input, cf_t* output) { ofdm_run(q, input, output, 0, 0, false); } // ofdm.c:384-393

static bool mbsfn_ready(srslte_ofdm_t* q)
{
  if (q->mbsfn_subframe) return true;
  ERROR("MBSFN slot call on an object that was not initialised with srslte_ofdm_%s_init_mbsfn", q->fft_plan.forward ? "rx" : "tx");
  return false;
}
void srslte_ofdm_rx_slot_mbsfn(srslte_ofdm_t* q, cf_t* input, cf_t* output)
{ // ofdm.c:424-437
  if (mbsfn_ready(q)) ofdm_run(q, input, output, 0, 0, true);
}
void srslte_ofdm_tx_slot_mbsfn(srslte_ofdm_t* q, cf_t* input, cf_t* output)
{ // ofdm.c:558-574
  if (mbsfn_ready(q)) ofdm_run(q, input, output, 0, 0, true);
}

// ====================================================================================================== transform precoding
bool srslte_dft_precoding_valid_prb(uint32_t nof_prb) { return srslte_hip_dft_precoding_valid_prb(nof_prb) != 0; }

int srslte_dft_precoding_init(srslte_dft_precoding_t* q, uint32_t max_prb, bool is_tx)
{ // dft_precoding.c:39-69: one normalised plan per valid nof_prb
  if (!q || max_prb > SRSLTE_MAX_PRB) return SRSLTE_ERROR_INVALID_INPUTS;
  memset(q, 0, sizeof(*q));
  for (uint32_t i = 1; i <= max_prb; i++) {
    if (srslte_dft_precoding_valid_prb(i)) {
      if (srslte_dft_plan_c(&q->dft_plan[i], 12 * (int)i, is_tx ? SRSLTE_DFT_FORWARD : SRSLTE_DFT_BACKWARD)) {
        srslte_dft_precoding_free(q);
        return SRSLTE_ERROR;
      }
      srslte_dft_plan_set_norm(&q->dft_plan[i], true);
    }
  }
  q->max_prb = max_prb;
  return SRSLTE_SUCCESS;
}
int  srslte_dft_precoding_init_tx(srslte_dft_precoding_t* q, uint32_t max_prb) { return srslte_dft_precoding_init(q, max_prb, true); }
int  srslte_dft_precoding_init_rx(srslte_dft_precoding_t* q, uint32_t max_prb) { return srslte_dft_precoding_init(q, max_prb, false); }
void srslte_dft_precoding_free(srslte_dft_precoding_t* q)
{
  for (uint32_t i = 1; i <= q->max_prb && i <= SRSLTE_MAX_PRB; i++) {
    if (srslte_dft_precoding_valid_prb(i)) srslte_dft_plan_free(&q->dft_plan[i]);
  }
  memset(q, 0, sizeof(*q));
}

int srslte_dft_precoding(srslte_dft_precoding_t* q, cf_t* input, cf_t* output, uint32_t nof_prb, uint32_t nof_symbols)
{ // dft_precoding.c:100-113
  if (!srslte_dft_precoding_valid_prb(nof_prb) || nof_prb > q->max_prb) {
    ERROR("Error invalid number of PRB (%u)", nof_prb);
    return SRSLTE_ERROR;
  }
  srslte_dft_plan_t* plan = &q->dft_plan[nof_prb];
  auto*              st   = (DftState*)plan->p;
  const int          N    = 12 * (int)nof_prb;
  const size_t       n    = sizeof(cf_t) * (size_t)N * nof_symbols;
  void *             di = st->in.get(n), *dout = st->out.get(n);
  if (!di || !dout || !h2d(di, input, n)) return SRSLTE_ERROR;
  if (srslte_hip_dft_precoding_batch(di, dout, nof_prb, nof_symbols, plan->forward ? 1 : 0, tl_stream())) return SRSLTE_ERROR;
  return d2h(output, dout, n) ? SRSLTE_SUCCESS : SRSLTE_ERROR;
}

// ====================================================================================================== segmentation / interleaver
int  srslte_cbsegm(srslte_cbsegm_t* s, uint32_t tbs) { return srslte_hip_cbsegm((srslte_hip_cbsegm_t*)s, tbs); }
int  srslte_cbsegm_cbsize(uint32_t index) { return srslte_hip_cbsegm_cbsize(index); }
int  srslte_cbsegm_cbindex(uint32_t long_cb) { return srslte_hip_cbsegm_cbindex(long_cb); }
bool srslte_cbsegm_cbsize_isvalid(uint32_t size)
{
  const int i = lte_cb_index(size);
  return i >= 0 && lte_qpp_table[i].K == size;
}

// srslte_tc_interl_init / _free: compat_refsignal.cpp, next to the 25.212 generator they were written for (tc_interl_umts.c:58-78)
int srslte_tc_interl_LTE_gen_interl(srslte_tc_interl_t* h, uint32_t long_cb, uint32_t interl_win)
{ // tc_interl_lte.c:75-114
  if (long_cb > h->max_long_cb) {
    ERROR("Interleaver initiated for max_long_cb=%u", h->max_long_cb);
    return SRSLTE_ERROR;
  }
  return srslte_hip_tc_interl_LTE_gen_interl(h->forward, h->reverse, long_cb, interl_win);
}
int srslte_tc_interl_LTE_gen(srslte_tc_interl_t* h, uint32_t long_cb) { return srslte_tc_interl_LTE_gen_interl(h, long_cb, 1); }

// ====================================================================================================== turbo encoder
int srslte_tcod_init(srslte_tcod_t* h, uint32_t max_long_cb)
{ // turbocoder.c:49-60
  h->max_long_cb = max_long_cb;
  h->temp        = (uint8_t*)host_alloc(max_long_cb / 8 + 1);
  return h->temp ? SRSLTE_SUCCESS : SRSLTE_ERROR;
}
void srslte_tcod_free(srslte_tcod_t* h)
{
  h->max_long_cb = 0;
  free(h->temp);
  h->temp = nullptr;
}
int srslte_tcod_encode(srslte_tcod_t* h, uint8_t* input, uint8_t* output, uint32_t long_cb)
{ // turbocoder.c:76-186
  if (long_cb > h->max_long_cb) {
    ERROR("Turbo coder initiated for max_long_cb=%u", h->max_long_cb);
    return SRSLTE_ERROR;
  }
  void *di = g_tcod_in.get(long_cb), *dout = g_tcod_out.get(3 * long_cb + 12);
  if (!di || !dout || !h2d(di, input, long_cb)) return SRSLTE_ERROR;
  if (srslte_hip_tcod_encode_batch((const uint8_t*)di, (uint8_t*)dout, long_cb, 1, tl_stream())) return SRSLTE_ERROR;
  return d2h(output, dout, 3 * long_cb + 12) ? SRSLTE_SUCCESS : SRSLTE_ERROR;
}

void srslte_tcod_gentable(void) {} // turbocoder.c:369-425 builds host LUTs; the device tables are built on first use per K

static inline void crc_put_byte(srslte_crc_t* h, uint8_t byte)
{ // crc.h:67-75
  const int ord = h->order - 8;
  uint64_t  crc = h->crcinit;
  h->crcinit    = (crc << 8) ^ h->table[((crc >> ord) & 0xff) ^ byte];
}

int srslte_tcod_encode_lut(srslte_tcod_t* h, srslte_crc_t* crc_tb, srslte_crc_t* crc_cb, uint8_t* input, uint8_t* parity, uint32_t cblen_idx,
                           bool last_cb)
{ // turbocoder.c:189-367: CRC attachment on the host exactly as upstream (running TB checksum across calls), encoder on the device
  if (cblen_idx >= 188) return SRSLTE_ERROR;
  const uint32_t long_cb = (uint32_t)lte_qpp_table[cblen_idx].K, nbytes = long_cb / 8;
  if (h && long_cb > h->max_long_cb) {
    ERROR("Turbo coder initiated for max_long_cb=%u", h->max_long_cb);
    return SRSLTE_ERROR;
  }
  if (!crc_tb || long_cb < (uint32_t)((crc_cb ? crc_cb->order : 0) + (last_cb ? crc_tb->order : 0))) return SRSLTE_ERROR_INVALID_INPUTS;
  auto append = [&](srslte_crc_t* c, uint32_t at, bool into_cb) { // :231-258 / :279-290: checksum bytes MSB first
    const uint32_t checksum = (uint32_t)(c->crcinit & c->crcmask);
    for (int i = 0; i < c->order / 8; i++) {
      const uint8_t in = (uint8_t)((checksum >> (8 * (c->order / 8 - i - 1))) & 0xff);
      if (into_cb) crc_put_byte(crc_cb, in);
      input[at + i] = in;
    }
  };
  if (crc_cb) {
    crc_cb->crcinit = 0; // srslte_crc_set_init(crc_cb, 0), :207-209
    const uint32_t n = (long_cb - crc_cb->order - (last_cb ? crc_tb->order : 0)) / 8;
    for (uint32_t i = 0; i < n; i++) {
      crc_put_byte(crc_tb, input[i]);
      crc_put_byte(crc_cb, input[i]);
    }
    if (last_cb) append(crc_tb, n, true);
    append(crc_cb, (long_cb - crc_cb->order) / 8, false);
  } else {
    const uint32_t n = (long_cb - (last_cb ? crc_tb->order : 0)) / 8;
    for (uint32_t i = 0; i < n; i++) crc_put_byte(crc_tb, input[i]);
    if (last_cb) append(crc_tb, n, false);
  }
  const uint32_t npar = long_cb / 4 + 1;
  const uint8_t* di   = zc_in(input, nbytes);
  uint8_t *      dpar = zc_out(parity, npar), *dtail = zc_out(&input[nbytes], 1);
  if (!di || !dpar || !dtail) return SRSLTE_ERROR;
  if (srslte_hip_tcod_encode_bytes_batch(di, nbytes, dpar, npar, dtail, long_cb, 1, tl_stream())) {
    (void)g_link.flush();
    return SRSLTE_ERROR;
  }
  if (!g_link.flush()) return SRSLTE_ERROR;
  return (int)(3 * long_cb + 12);
}

// ====================================================================================================== turbo decoder
uint32_t srslte_tdec_autoimp_get_subblocks(uint32_t long_cb) { return srslte_hip_tdec_autoimp_get_subblocks(long_cb); }
uint32_t srslte_tdec_autoimp_get_subblocks_8bit(uint32_t long_cb) { return srslte_hip_tdec_autoimp_get_subblocks_8bit(long_cb); }

int srslte_tdec_init_manual(srslte_tdec_t* h, uint32_t max_long_cb, srslte_tdec_impl_type_t dec_type)
{ // turbodecoder.c:165-330; the device object replaces app/ext/beta work buffers
  if (!h) return SRSLTE_ERROR_INVALID_INPUTS;
  memset(h, 0, sizeof(*h));
  switch (dec_type) {
    case SRSLTE_TDEC_AUTO:
    case SRSLTE_TDEC_GENERIC:
    case SRSLTE_TDEC_SSE_WINDOW:
    case SRSLTE_TDEC_AVX_WINDOW:
    case SRSLTE_TDEC_SSE8_WINDOW:
    case SRSLTE_TDEC_AVX8_WINDOW: break;
    default: ERROR("Error decoder %d not supported", (int)dec_type); return SRSLTE_ERROR;
  }
  auto* st = new TdecState();
  st->h    = srslte_hip_tdec_create(max_long_cb, 1);
  if (!st->h) {
    delete st;
    return SRSLTE_ERROR;
  }
  h->dec16_hdlr[0]    = st;
  h->max_long_cb      = max_long_cb;
  h->dec_type         = dec_type;
  h->current_llr_type = dec_type >= SRSLTE_TDEC_SSE8_WINDOW ? SRSLTE_TDEC_8 : SRSLTE_TDEC_16;
  h->current_cbidx    = -1;
  return SRSLTE_SUCCESS;
}
int srslte_tdec_init(srslte_tdec_t* h, uint32_t max_long_cb) { return srslte_tdec_init_manual(h, max_long_cb, SRSLTE_TDEC_AUTO); }

void srslte_tdec_free(srslte_tdec_t* h)
{
  auto* st = (TdecState*)h->dec16_hdlr[0];
  if (st) {
    srslte_hip_tdec_destroy(st->h);
    srslte_hip_sch_destroy(st->sch);
    st->in.release();
    st->out.release();
    delete st;
  }
  memset(h, 0, sizeof(*h));
}
void srslte_tdec_force_not_sb(srslte_tdec_t* h) { h->force_not_sb = true; }
int  srslte_tdec_get_nof_iterations(srslte_tdec_t* h) { return h->n_iter; }

int srslte_tdec_new_cb(srslte_tdec_t* h, uint32_t long_cb)
{ // turbodecoder.c:522-537
  if (long_cb > h->max_long_cb) {
    ERROR("TDEC was initialized for max_long_cb=%u", h->max_long_cb);
    return SRSLTE_ERROR;
  }
  h->n_iter          = 0;
  h->current_long_cb = long_cb;
  h->current_cbidx   = srslte_cbsegm_cbindex(long_cb);
  if (h->current_cbidx < 0 || lte_qpp_table[h->current_cbidx].K != long_cb) {
    ERROR("Invalid CB length %u", long_cb);
    h->current_cbidx = -1;
    return SRSLTE_ERROR;
  }
  return SRSLTE_SUCCESS;
}

// One call = `passes` SISO passes from the unchanged input. The back-end follows turbodecoder.c:438-520: AUTO picks per K
// and per LLR width; a manual type fixes width and window count, and the other API width is converted with a C cast
// (convert_8_to_16 / convert_16_to_8, :451-463).
static int tdec_passes(srslte_tdec_t* h, const void* input, bool api8, uint8_t* output, uint32_t passes, uint32_t start = 0)
{
  g_stats[2]++; // start > 0: passes 0..start-1 were run by the previous call on this object for this code block; only the rest is run
  auto*          st = (TdecState*)h->dec16_hdlr[0];
  const uint32_t K  = h->current_long_cb;
  int            W  = -1;   // AUTO
  bool           dec8 = api8;
  switch (h->dec_type) {
    case SRSLTE_TDEC_GENERIC: W = 0; dec8 = false; break;
    case SRSLTE_TDEC_SSE_WINDOW: W = 8; dec8 = false; break;
    case SRSLTE_TDEC_AVX_WINDOW: W = 16; dec8 = false; break;
    case SRSLTE_TDEC_SSE8_WINDOW: W = 16; dec8 = true; break;
    case SRSLTE_TDEC_AVX8_WINDOW: W = 32; dec8 = true; break;
    default: break;
  }
  const uint32_t nsb = W >= 0 ? (uint32_t)W : (api8 ? srslte_hip_tdec_autoimp_get_subblocks_8bit(K) : srslte_hip_tdec_autoimp_get_subblocks(K));
  // SB input layout (turbodecoder_iter.h:84 input_is_interleaved): AUTO window back-ends and every 8-bit back-end
  const bool interleaved = W < 0 ? nsb > 0 : dec8;
  const int  sb  = (!h->force_not_sb && interleaved && nsb > 0) ? 1 : 0;
  const uint32_t len = srslte_hip_tdec_input_len(K, sb);
  const size_t   esz = dec8 ? 1 : 2;
  void *         di = st->in.get(len * esz), *dout = st->out.get(K / 8);
  if (!di || !dout) return SRSLTE_ERROR;
  if (start > 0) {
    // later passes of the same block: upstream reads the input on pass 0 only (extract_input, turbodecoder_iter.h:84-99); the device copy
    // of this object is still there
  } else if (dec8 == api8) {
    if (!h2d(di, input, len * esz)) return SRSLTE_ERROR;
  } else if (dec8) { // 16-bit API on a manual 8-bit back-end
    std::vector<int8_t> c(len);
    for (uint32_t i = 0; i < len; i++) c[i] = (int8_t)((const int16_t*)input)[i];
    if (!h2d(di, c.data(), len)) return SRSLTE_ERROR;
  } else { // 8-bit API on a manual 16-bit back-end
    std::vector<int16_t> c(len);
    for (uint32_t i = 0; i < len; i++) c[i] = ((const int8_t*)input)[i];
    if (!h2d(di, c.data(), len * 2)) return SRSLTE_ERROR;
  }
  tdec_set_resume(st->h, start);
  if (tdec_run_batch_w(st->h, di, dec8 ? 1 : 0, len, sb, K, W, 1, passes, 0, 0, (uint8_t*)dout, K / 8, nullptr, nullptr, (hipStream_t)tl_stream()))
    return SRSLTE_ERROR;
  return d2h(output, dout, K / 8) ? SRSLTE_SUCCESS : SRSLTE_ERROR;
}

static void tdec_one_more(srslte_tdec_t* h, const void* input, bool api8, uint8_t* output)
{ // turbodecoder.c:539-545,:565-571. One more SISO pass: the decoder's work arrays for this object's block slot stay on the device
  // between calls, so pass n_iter continues from them; the input went up with pass 0.
  if (h->current_cbidx >= 0) {
    if (tdec_passes(h, input, api8, output, (uint32_t)h->n_iter + 1, (uint32_t)h->n_iter) == SRSLTE_SUCCESS) h->n_iter++;
  } else {
    ERROR("Error CB index not set (call srslte_tdec_new_cb() first");
  }
}
void srslte_tdec_iteration(srslte_tdec_t* h, int16_t* input, uint8_t* output) { tdec_one_more(h, input, false, output); }
void srslte_tdec_iteration_8bit(srslte_tdec_t* h, int8_t* input, uint8_t* output) { tdec_one_more(h, input, true, output); }

// ---- srslte_dlsch_decode2 (sch.c:507-531) = decode_tb (sch.c:429-500) with all code blocks in one device call (srslte_hip_sch_decode)
int srslte_dlsch_decode2(srslte_sch_t* q, srslte_pdsch_cfg_t* cfg, int16_t* e_bits, uint8_t* data, int tb_idx, uint32_t nof_layers)
{
  if (!q || !cfg || tb_idx < 0 || tb_idx >= SRSLTE_MAX_CODEWORDS) return SRSLTE_ERROR_INVALID_INPUTS;
  const uint32_t   Nl = nof_layers != cfg->grant.nof_tb ? 2 : 1; // :510-514
  srslte_cbsegm_t  seg;
  const srslte_ra_tb_t& tb = cfg->grant.tb[tb_idx];
  if (srslte_cbsegm(&seg, (uint32_t)tb.tbs)) {
    ERROR("Error computing Codeword (%d) segmentation for TBS=%d", tb_idx, tb.tbs);
    return SRSLTE_ERROR;
  }
  srslte_softbuffer_rx_t* sb = cfg->softbuffers.rx[tb_idx];
  if (!data || !sb || !e_bits) { // decode_tb :437-441,:491-498
    ERROR("Missing inputs: data=%d, softbuffer=%d, e_bits=%d", data != 0, sb != 0, e_bits != 0);
    return SRSLTE_ERROR_INVALID_INPUTS;
  }
  if (seg.tbs == 0 || seg.C == 0) return SRSLTE_SUCCESS; // :444-446
  g_stats[1]++;
  if (seg.F) {
    fprintf(stderr, "Error filler bits are not supported. Use standard TBS\n"); // :448-451
    return SRSLTE_ERROR_INVALID_INPUTS;
  }
  if (seg.C > sb->max_cb) {
    fprintf(stderr, "Error number of CB to decode (%d) exceeds soft buffer size (%d CBs)\n", seg.C, sb->max_cb); // :453-457
    return SRSLTE_ERROR_INVALID_INPUTS;
  }
  auto* st = (TdecState*)q->decoder.dec16_hdlr[0];
  if (!st || tb.mod < SRSLTE_MOD_QPSK || tb.mod > SRSLTE_MOD_256QAM || tb.nof_bits == 0) return SRSLTE_ERROR_INVALID_INPUTS;
  if (!st->sch || (uint32_t)tb.tbs > st->sch_tbs || tb.nof_bits > st->sch_e || st->sch_l8 != q->llr_is_8bit) {
    srslte_hip_sch_destroy(st->sch);
    st->sch_tbs = (uint32_t)tb.tbs > st->sch_tbs ? (uint32_t)tb.tbs : st->sch_tbs;
    st->sch_e   = tb.nof_bits > st->sch_e ? tb.nof_bits : st->sch_e;
    st->sch_l8  = q->llr_is_8bit;
    st->sch     = srslte_hip_sch_create(st->sch_tbs, st->sch_e, st->sch_l8 ? 1 : 0);
    if (!st->sch) {
      st->sch_tbs = st->sch_e = 0;
      return SRSLTE_ERROR;
    }
  }
  const uint32_t C = seg.C, tbs8 = seg.tbs / 8;
  data[tbs8 + 0] = data[tbs8 + 1] = data[tbs8 + 2] = 0; // :461-463
  st->cb_bytes.resize((size_t)C * 768);
  st->cb_crc.resize(C);
  for (uint32_t i = 0; i < C; i++) st->cb_crc[i] = sb->cb_crc[i] ? 1 : 0;
  uint32_t passes = 0;
  const int r = srslte_hip_sch_decode(st->sch, e_bits, tb.nof_bits, (uint32_t)tb.tbs, (int)tb.mod, Nl, (uint32_t)tb.rv, q->max_iterations, sb->buffer_f,
                                      st->cb_crc.data(), st->cb_bytes.data(), &passes);
  if (r) return r == SRSLTE_ERROR_INVALID_INPUTS ? SRSLTE_ERROR_INVALID_INPUTS : SRSLTE_ERROR;
  // decode_tb_cb :312-412: bytes of every block in order (a decoded block's K / 8 bytes start where its rlen / 8 payload bytes go; the next
  // block overwrites its 24 CRC bits), blocks decoded in an earlier transmission from the soft buffer's copy
  for (uint32_t i = 0; i < C; i++) {
    const uint32_t K = i < seg.C1 ? seg.K1 : seg.K2, rlen = C == 1 ? K : K - 24;
    if (sb->cb_crc[i]) {
      memcpy(&data[(size_t)i * rlen / 8], sb->data[i], rlen / 8);
    } else {
      memcpy(&data[(size_t)i * rlen / 8], &st->cb_bytes[(size_t)i * 768], K / 8);
      sb->cb_crc[i] = st->cb_crc[i] != 0;
    }
  }
  q->avg_iterations = (float)passes / (float)C; // :313,:361,:412
  sb->tb_crc = true;
  for (uint32_t i = 0; i < C && sb->tb_crc; i++) sb->tb_crc = sb->cb_crc[i];
  if (!sb->tb_crc) { // save the blocks that passed for the next transmission (:400-410)
    for (uint32_t i = 0; i < C; i++) {
      const uint32_t K = i < seg.C1 ? seg.K1 : seg.K2, rlen = C == 1 ? K : K - 24;
      if (sb->cb_crc[i]) memcpy(sb->data[i], &data[(size_t)i * rlen / 8], rlen / 8);
    }
    return SRSLTE_ERROR;
  }
  // transport block CRC24A over the payload against the three bytes behind it (:470-488), with the table of q->crc_tb (crc.c:139-153)
  srslte_crc_t* c = &q->crc_tb;
  c->crcinit      = 0;
  for (uint32_t i = 0; i < tbs8; i++) crc_put_byte(c, data[i]);
  const uint32_t par_rx = (uint32_t)(c->crcinit & c->crcmask);
  const uint32_t par_tx = ((uint32_t)data[tbs8] << 16) | ((uint32_t)data[tbs8 + 1] << 8) | (uint32_t)data[tbs8 + 2];
  return (par_rx == par_tx && par_rx) ? SRSLTE_SUCCESS : SRSLTE_ERROR;
}
int srslte_dlsch_decode(srslte_sch_t* q, srslte_pdsch_cfg_t* cfg, int16_t* e_bits, uint8_t* data) { return srslte_dlsch_decode2(q, cfg, e_bits, data, 0, 1); }

static int tdec_all(srslte_tdec_t* h, const void* input, bool api8, uint8_t* output, uint32_t nof_iterations, uint32_t long_cb)
{ // turbodecoder.c:547-562,:573-588
  if (srslte_tdec_new_cb(h, long_cb)) return SRSLTE_ERROR;
  if (nof_iterations == 0) nof_iterations = 1; // do { } while: at least one pass
  if (tdec_passes(h, input, api8, output, nof_iterations)) return SRSLTE_ERROR;
  h->n_iter = (int)nof_iterations;
  return SRSLTE_SUCCESS;
}
int srslte_tdec_run_all(srslte_tdec_t* h, int16_t* input, uint8_t* output, uint32_t nof_iterations, uint32_t long_cb)
{
  return tdec_all(h, input, false, output, nof_iterations, long_cb);
}
int srslte_tdec_run_all_8bit(srslte_tdec_t* h, int8_t* input, uint8_t* output, uint32_t nof_iterations, uint32_t long_cb)
{
  return tdec_all(h, input, true, output, nof_iterations, long_cb);
}

// ====================================================================================================== channel estimator
int srslte_chest_dl_init(srslte_chest_dl_t* q, uint32_t max_prb, uint32_t nof_rx_antennas)
{ // chest_dl.c:69-160
  if (!q || nof_rx_antennas == 0 || nof_rx_antennas > SRSLTE_MAX_PORTS || max_prb > SRSLTE_MAX_PRB) return SRSLTE_ERROR_INVALID_INPUTS;
  memset(q, 0, sizeof(*q));
  q->nof_rx_antennas = nof_rx_antennas;
  if (srslte_refsignal_cs_init(&q->csr_refs, max_prb)) return SRSLTE_ERROR; // the host copy callers read (chest_dl.c:88, chest_test_dl.c:154)
  q->tmp_noise = (cf_t*)new ChestState(); // opaque slot for the device state
  return SRSLTE_SUCCESS;
}

static void chest_drop_cell(srslte_chest_dl_t* q)
{
  auto* st = (ChestState*)q->tmp_noise;
  if (st && st->h) {
    srslte_hip_chest_dl_destroy(st->h);
    st->h = nullptr;
  }
}

void srslte_chest_dl_free(srslte_chest_dl_t* q)
{ // chest_dl.c:162-191
  if (!q) return;
  chest_drop_cell(q);
  srslte_refsignal_free(&q->csr_refs);
  auto* st = (ChestState*)q->tmp_noise;
  if (st) {
    st->grid.release();
    st->ce.release();
    st->res.release();
    delete st;
  }
  memset(q, 0, sizeof(*q));
}

int srslte_chest_dl_set_cell(srslte_chest_dl_t* q, srslte_cell_t cell)
{ // chest_dl.c:244-300
  if (!q || !q->tmp_noise || cell.nof_prb < 6 || cell.nof_prb > SRSLTE_MAX_PRB || cell.id > 503) return SRSLTE_ERROR_INVALID_INPUTS;
  if (q->cell.id == cell.id && q->cell.nof_prb == cell.nof_prb && ((ChestState*)q->tmp_noise)->h) return SRSLTE_SUCCESS;
  chest_drop_cell(q);
  auto* st = (ChestState*)q->tmp_noise;
  st->h    = srslte_hip_chest_dl_create(cell.id, cell.nof_prb, cell.nof_ports, cell.cp == SRSLTE_CP_NORM);
  if (!st->h) return SRSLTE_ERROR;
  q->cell = cell;
  // host copy of the CRS values for callers that use srslte_refsignal_cs_put_sf(&q->csr_refs, ...) (chest_test_dl.c:154); the device
  // object was rebuilt, so is this (upstream's table would survive a change of width alone, refsignal_dl.c:77)
  srslte_refsignal_free(&q->csr_refs);
  if (srslte_refsignal_cs_init(&q->csr_refs, cell.nof_prb)) return SRSLTE_ERROR;
  return srslte_refsignal_cs_set_cell(&q->csr_refs, cell) ? SRSLTE_ERROR : SRSLTE_SUCCESS;
}

int srslte_chest_dl_set_mbsfn_area_id(srslte_chest_dl_t* q, uint16_t mbsfn_area_id)
{ // chest_dl.c:244-262: builds the area's MBSFN reference signal (on the device)
  auto* st = q ? (ChestState*)q->tmp_noise : nullptr;
  if (!st || !st->h) return SRSLTE_ERROR_INVALID_INPUTS;
  if (mbsfn_area_id >= 256) return SRSLTE_ERROR; // SRSLTE_MAX_MBSFN_AREA_IDS; // upstream returns -1 without a message (:260)
  return srslte_hip_chest_dl_set_mbsfn_area_id(st->h, mbsfn_area_id) ? SRSLTE_ERROR : SRSLTE_SUCCESS;
}

int srslte_chest_dl_res_init(srslte_chest_dl_res_t* q, uint32_t max_prb)
{ // chest_dl.c:193-210
  memset(q, 0, sizeof(*q));
  q->nof_re = 14 * 12 * max_prb;
  for (int i = 0; i < SRSLTE_MAX_PORTS; i++) {
    for (int j = 0; j < SRSLTE_MAX_PORTS; j++) {
      q->ce[i][j] = (cf_t*)host_alloc(sizeof(cf_t) * q->nof_re);
      if (!q->ce[i][j]) return SRSLTE_ERROR;
      bzero(q->ce[i][j], sizeof(cf_t) * q->nof_re);
    }
  }
  return SRSLTE_SUCCESS;
}
static void res_fill(srslte_chest_dl_res_t* q, bool identity)
{
  for (int i = 0; i < SRSLTE_MAX_PORTS; i++) {
    for (int j = 0; j < SRSLTE_MAX_PORTS; j++) {
      float* p = (float*)q->ce[i][j];
      if (!p) continue;
      for (uint32_t k = 0; k < q->nof_re; k++) {
        p[2 * k]     = (!identity || i == j) ? 1.0f : 0.0f;
        p[2 * k + 1] = 0.0f;
      }
    }
  }
}
void srslte_chest_dl_res_set_identity(srslte_chest_dl_res_t* q) { res_fill(q, true); } // chest_dl.c:212-221
void srslte_chest_dl_res_set_ones(srslte_chest_dl_res_t* q) { res_fill(q, false); }    // chest_dl.c:223-231
void srslte_chest_dl_res_free(srslte_chest_dl_res_t* q)
{
  for (int i = 0; i < SRSLTE_MAX_PORTS; i++) {
    for (int j = 0; j < SRSLTE_MAX_PORTS; j++) free(q->ce[i][j]);
  }
  memset(q, 0, sizeof(*q));
}

// MBSFN subframes (chest_dl.c:718-745,:892-896): estimates and, with the REFS algorithm, the noise come from the device; rsrp, rssi, cfo,
// the sync error (and the noise with PSS / EMPTY) are what the last normal subframe left in q, and fill_res (:845-871) reads those
static int chest_dl_estimate_mbsfn(srslte_chest_dl_t* q, ChestState* st, srslte_dl_sf_cfg_t* sf, srslte_chest_dl_cfg_t* cfg,
                                   cf_t* input[SRSLTE_MAX_PORTS], srslte_chest_dl_res_t* res)
{
  const uint32_t nrx = q->nof_rx_antennas, npt = q->cell.nof_ports;
  const size_t   n   = sizeof(cf_t) * 2 * cp_nsymb(q->cell.cp) * 12 * q->cell.nof_prb; // SRSLTE_SF_LEN_RE
  char *         dg = (char*)st->grid.get(n * nrx), *dce = (char*)st->ce.get(n * nrx * npt);
  float*         dnoise = (float*)st->res.get(sizeof(float) * 16);
  if (!dg || !dce || !dnoise) return SRSLTE_ERROR;
  bool want_ce = false;
  for (uint32_t a = 0; a < nrx; a++) {
    if (!input[a] || !h2d(dg + a * n, input[a], n)) return SRSLTE_ERROR_INVALID_INPUTS;
    for (uint32_t pt = 0; pt < npt; pt++) want_ce = want_ce || res->ce[pt][a];
  }
  if (want_ce) { // the caller's symbols 12, 13 (not part of the 12-symbol subframe) stay as they are
    for (uint32_t pt = 0; pt < npt; pt++) {
      for (uint32_t a = 0; a < nrx; a++) {
        if (res->ce[pt][a] && !h2d(dce + (pt * nrx + a) * n, res->ce[pt][a], n)) return SRSLTE_ERROR;
      }
    }
  }
  srslte_hip_chest_dl_cfg_t hc;
  memset(&hc, 0, sizeof(hc));
  hc.noise_alg = cfg->noise_alg; hc.filter_type = cfg->filter_type; hc.filter_coef[0] = cfg->filter_coef[0]; hc.filter_coef[1] = cfg->filter_coef[1];
  hc.interpolate_subframe = cfg->interpolate_subframe; hc.mbsfn_area_id = cfg->mbsfn_area_id;
  srslte_hip_chest_dl_set_symbol_sz(st->h, srslte_symbol_sz(q->cell.nof_prb)); // chest_dl.c:575,:695: read at every call
  if (srslte_hip_chest_dl_estimate_mbsfn_batch(st->h, &hc, sf->tti % 10, dg, want_ce ? dce : nullptr, dnoise, 1, (int)nrx, tl_stream())) return SRSLTE_ERROR;
  if (cfg->noise_alg == SRSLTE_NOISE_ALG_REFS) {
    float nz[16];
    if (!d2h(nz, dnoise, sizeof(float) * nrx * npt)) return SRSLTE_ERROR;
    for (uint32_t pt = 0; pt < npt; pt++) {
      for (uint32_t a = 0; a < nrx; a++) q->noise_estimate[a][pt] = nz[pt * nrx + a];
    }
  }
  for (uint32_t pt = 0; pt < npt; pt++) {
    for (uint32_t a = 0; a < nrx; a++) {
      if (res->ce[pt][a] && !d2h(res->ce[pt][a], dce + (pt * nrx + a) * n, n)) return SRSLTE_ERROR;
    }
  }
  // fill_res (chest_dl.c:747-871) on the estimator's kept state
  auto  dbm = [](float a) { return (float)(10 * log10(a) + 30); };
  auto  db  = [](float a) { return (float)(10 * log10(a)); };
  float noise = 0.f, rssi = 0.f, rsrq = 0.f, rsrp = -1e9f, neigh = -1e9f;
  for (uint32_t a = 0; a < nrx; a++) {
    float s = 0.f, c = 0.f;
    for (uint32_t pt = 0; pt < npt; pt++) {
      s += q->noise_estimate[a][pt];
      c += q->rsrp_corr[a][pt];
    }
    noise += s / npt;
    rssi += 4 * q->rssi[a][0] / q->cell.nof_prb / 12;
    rsrq += q->cell.nof_prb * q->rsrp[a][0] / q->rssi[a][0];
    float v = 0.f; // get_rsrp indexes the ports with the antenna counter (:809-819)
    for (uint32_t j = 0; j < nrx; j++) v += q->rsrp[j][a];
    v /= nrx;
    rsrp  = v > rsrp ? v : rsrp;
    neigh = c / npt > neigh ? c / npt : neigh;
  }
  noise /= nrx; rssi /= nrx; rsrq /= nrx;
  res->noise_estimate = noise; res->noise_estimate_dbm = dbm(noise); res->cfo = q->cfo; res->rsrp = rsrp; res->rsrp_dbm = dbm(rsrp);
  res->rsrp_neigh = neigh; res->rsrq = rsrq; res->rsrq_db = db(rsrq); res->snr_db = db(rsrp / noise); res->rssi_dbm = dbm(rssi);
  res->sync_error = q->sync_err[0][0];
  for (uint32_t pt = 0; pt < npt; pt++) {
    float mean_rsrp = 0.f;
    for (uint32_t a = 0; a < nrx; a++) {
      mean_rsrp += q->rsrp[a][pt] / nrx;
      res->snr_ant_port_db[a][pt]   = db(q->rsrp[a][pt] / q->noise_estimate[a][pt]);
      res->rsrp_ant_port_dbm[a][pt] = dbm(q->rsrp[a][pt]);
      res->rsrq_ant_port_db[a][pt]  = db(q->cell.nof_prb * q->rsrp[a][pt] / q->rssi[a][pt]);
    }
    res->rsrp_port_dbm[pt] = dbm(mean_rsrp);
  }
  return SRSLTE_SUCCESS;
}

int srslte_chest_dl_estimate_cfg(srslte_chest_dl_t* q, srslte_dl_sf_cfg_t* sf, srslte_chest_dl_cfg_t* cfg, cf_t* input[SRSLTE_MAX_PORTS],
                                 srslte_chest_dl_res_t* res)
{ // chest_dl.c:884-908
  auto* st = q ? (ChestState*)q->tmp_noise : nullptr;
  if (!st || !st->h || !sf || !cfg || !input || !res) return SRSLTE_ERROR_INVALID_INPUTS;
  if (sf->sf_type == SRSLTE_SF_MBSFN) return chest_dl_estimate_mbsfn(q, st, sf, cfg, input, res);
  const uint32_t nrx = q->nof_rx_antennas, npt = q->cell.nof_ports;
  const size_t   n   = sizeof(cf_t) * 2 * cp_nsymb(q->cell.cp) * 12 * q->cell.nof_prb; // SRSLTE_SF_LEN_RE
  char *         dg = (char*)st->grid.get(n * nrx), *dce = (char*)st->ce.get(n * nrx * npt);
  void*          dres = st->res.get(sizeof(srslte_hip_chest_dl_res_t));
  if (!dg || !dce || !dres) return SRSLTE_ERROR;
  bool want_ce = false;
  for (uint32_t a = 0; a < nrx; a++) {
    if (!input[a] || !h2d(dg + a * n, input[a], n)) return SRSLTE_ERROR_INVALID_INPUTS;
    for (uint32_t pt = 0; pt < npt; pt++) want_ce = want_ce || res->ce[pt][a];
  }
  srslte_hip_chest_dl_cfg_t hc;
  memset(&hc, 0, sizeof(hc));
  hc.noise_alg = cfg->noise_alg; hc.filter_type = cfg->filter_type; hc.filter_coef[0] = cfg->filter_coef[0]; hc.filter_coef[1] = cfg->filter_coef[1];
  hc.interpolate_subframe = cfg->interpolate_subframe; hc.rsrp_neighbour = cfg->rsrp_neighbour;
  hc.cfo_estimate_enable  = cfg->cfo_estimate_enable && ((1u << (sf->tti % 10)) & cfg->cfo_estimate_sf_mask);
  hc.cfo_estimate_sf_mask = cfg->cfo_estimate_sf_mask; hc.sync_error_enable = cfg->sync_error_enable;
  if (hc.noise_alg != SRSLTE_NOISE_ALG_REFS) { // PSS / EMPTY renew q->noise_estimate in subframes 0 and 5 only (chest_dl.c:657-672)
    float state[16] = {0};
    for (uint32_t pt = 0; pt < npt; pt++) {
      for (uint32_t a = 0; a < nrx; a++) state[pt * nrx + a] = q->noise_estimate[a][pt];
    }
    if (chest_dl_set_noise_state(st->h, state)) return SRSLTE_ERROR;
  }
  if (want_ce && npt == 4 && hc.interpolate_subframe) {
    // ports 2/3: upstream replicates symbol 0 of the caller's estimates over the subframe (chest_dl.c:467-471, see chest_dl_kernel): that
    // symbol goes up so that the device has it
    for (uint32_t pt = 2; pt < npt; pt++) {
      for (uint32_t a = 0; a < nrx; a++) {
        if (res->ce[pt][a] && !h2d(dce + (pt * nrx + a) * n, res->ce[pt][a], sizeof(cf_t) * 12 * q->cell.nof_prb)) return SRSLTE_ERROR;
      }
    }
  }
  srslte_hip_chest_dl_set_symbol_sz(st->h, srslte_symbol_sz(q->cell.nof_prb)); // chest_dl.c:575,:695: read at every call
  // TDD cell: a special subframe has only the CRS symbols of its DwPTS (srslte_refsignal_cs_nof_symbols reads sf->tdd_config at every call)
  const bool tdd = q->cell.frame_type == SRSLTE_TDD && sf->tdd_config.configured && sf->tdd_config.sf_config < 7 && sf->tdd_config.ss_config < 10;
  if (srslte_hip_chest_dl_set_tdd(st->h, tdd ? (int)sf->tdd_config.sf_config : -1, tdd ? (int)sf->tdd_config.ss_config : 0)) return SRSLTE_ERROR;
  if (srslte_hip_chest_dl_estimate_batch_multi(st->h, &hc, sf->tti % 10, dg, want_ce ? dce : nullptr, dres, 1, (int)nrx, tl_stream())) return SRSLTE_ERROR;
  srslte_hip_chest_dl_res_t r;
  float raw[SRSLTE_MAX_PORTS * SRSLTE_MAX_PORTS][6]; // [port][antenna] {noise, rsrp, rssi, cfo, sync, corr}
  if (!d2h_later(&r, dres, sizeof(r))) return SRSLTE_ERROR;
  for (uint32_t pt = 0; pt < npt; pt++) {
    for (uint32_t a = 0; a < nrx; a++) {
      if (res->ce[pt][a] && !d2h_later(res->ce[pt][a], dce + (pt * nrx + a) * n, n)) return SRSLTE_ERROR;
    }
  }
  if (!d2h(raw, srslte_hip_chest_dl_last_raw(st->h), sizeof(float) * 6 * nrx * npt)) return SRSLTE_ERROR; // one wait for everything queued
  // fill_res, chest_dl.c:845-871
  if (hc.cfo_estimate_enable) q->cfo = r.cfo;
  q->sync_err[0][0]   = r.sync_error;
  res->noise_estimate = r.noise_estimate; res->noise_estimate_dbm = r.noise_estimate_dbm; res->snr_db = r.snr_db;
  res->rsrp = r.rsrp; res->rsrp_dbm = r.rsrp_dbm; res->rsrq = r.rsrq; res->rsrq_db = r.rsrq_db; res->rssi_dbm = r.rssi_dbm;
  res->cfo = q->cfo; res->sync_error = r.sync_error;
  if (hc.rsrp_neighbour) {
    for (uint32_t pt = 0; pt < npt; pt++) {
      for (uint32_t a = 0; a < nrx; a++) q->rsrp_corr[a][pt] = raw[pt * nrx + a][5];
    }
  }
  { // get_rsrp_neighbour (chest_dl.c:821-843): max over antennas of the port-mean of q->rsrp_corr (which keeps its last enabled values)
    float mx = -1e9f;
    for (uint32_t a = 0; a < nrx; a++) {
      float v = 0.f;
      for (uint32_t pt = 0; pt < npt; pt++) v += q->rsrp_corr[a][pt];
      v /= npt;
      mx = v > mx ? v : mx;
    }
    res->rsrp_neigh = mx;
  }
  if (nrx * npt == 1) {
    q->noise_estimate[0][0] = r.noise_estimate;
    q->rsrp[0][0]           = r.rsrp;
    q->rssi[0][0]           = raw[0][2];
    res->rsrp_port_dbm[0] = r.rsrp_dbm; res->snr_ant_port_db[0][0] = r.snr_db; res->rsrp_ant_port_dbm[0][0] = r.rsrp_dbm;
    res->rsrq_ant_port_db[0][0] = r.rsrq_db;
  } else { // per-antenna / per-port fields (chest_dl.c:860-870) from the per-(port, antenna) scalars the device kept
    for (uint32_t pt = 0; pt < npt; pt++) {
      float mean_rsrp = 0.f;
      for (uint32_t a = 0; a < nrx; a++) {
        const float* v = raw[pt * nrx + a];
        q->noise_estimate[a][pt] = v[0];
        q->rsrp[a][pt]           = v[1];
        q->rssi[a][pt]           = v[2];
        mean_rsrp += v[1] / nrx;
        res->snr_ant_port_db[a][pt]   = (float)(10 * log10(v[1] / v[0]));
        res->rsrp_ant_port_dbm[a][pt] = (float)(10 * log10(v[1]) + 30);
        res->rsrq_ant_port_db[a][pt]  = (float)(10 * log10(q->cell.nof_prb * v[1] / v[2]));
      }
      res->rsrp_port_dbm[pt] = (float)(10 * log10(mean_rsrp) + 30);
    }
  }
  return SRSLTE_SUCCESS;
}

int srslte_chest_dl_estimate(srslte_chest_dl_t* q, srslte_dl_sf_cfg_t* sf, cf_t* input[SRSLTE_MAX_PORTS], srslte_chest_dl_res_t* res)
{ // chest_dl.c:873-882: all-zero configuration
  srslte_chest_dl_cfg_t cfg;
  memset(&cfg, 0, sizeof(cfg));
  return srslte_chest_dl_estimate_cfg(q, sf, &cfg, input, res);
}

// ====================================================================================================== soft demapper
static int demod_host(int type, srslte_mod_t mod, const cf_t* symbols, void* llr, int nsymbols, size_t llr_elem)
{
  if ((int)mod < 0 || (int)mod > 4) {
    ERROR("Invalid modulation %d", (int)mod);
    return SRSLTE_ERROR;
  }
  if (nsymbols <= 0) return SRSLTE_SUCCESS;
  const int    Qm = mod == SRSLTE_MOD_BPSK ? 1 : 2 * (int)mod;
  const size_t nin = sizeof(cf_t) * nsymbols, nout = llr_elem * Qm * nsymbols;
  void *       di = g_demod_in.get(nin), *dout = g_demod_out.get(nout);
  if (!di || !dout || !h2d(di, symbols, nin)) return SRSLTE_ERROR;
  if (demod_launch(type, (int)mod, di, dout, nsymbols, 1, nullptr, 0, 0, (hipStream_t)tl_stream())) return SRSLTE_ERROR;
  return d2h(llr, dout, nout) ? SRSLTE_SUCCESS : SRSLTE_ERROR;
}
int srslte_demod_soft_demodulate(srslte_mod_t m, const cf_t* s, float* llr, int n) { return demod_host(0, m, s, llr, n, sizeof(float)); }
int srslte_demod_soft_demodulate_s(srslte_mod_t m, const cf_t* s, short* llr, int n) { return demod_host(1, m, s, llr, n, sizeof(short)); }
int srslte_demod_soft_demodulate_b(srslte_mod_t m, const cf_t* s, int8_t* llr, int n) { return demod_host(2, m, s, llr, n, sizeof(int8_t)); }

} // extern "C"
