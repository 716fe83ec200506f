// Device plumbing of the C ABI: lets a plain-C host (or ctypes) drive libsrslte_phy_hip.so without HIP headers.
#include "phy_hip_internal.hpp"
#include <stdarg.h>
#include <string.h>

extern "C" {
// the reference's logging hook, present only when the program also links lib/src/phy/utils/phy_logger.c (phy_logger.h:41-47, debug.h:46)
extern int handler_registered __attribute__((weak));
void       srslte_phy_log_print(int log_level, const char* format, ...) __attribute__((weak));
}

void hip_log(const char* fmt, ...)
{
  char    buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (&handler_registered && handler_registered && srslte_phy_log_print) {
    size_t n = strlen(buf);
    if (n && buf[n - 1] == '\n') buf[n - 1] = 0; // the reference's messages carry no newline when they go to a handler
    srslte_phy_log_print(2 /* LOG_LEVEL_ERROR_S */, "%s", buf);
  } else {
    fputs(buf, stderr);
  }
}

extern "C" int srslte_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int srslte_hip_set_device(int device)
{
  HIP_TRY(hipSetDevice(device));
  return SRSLTE_SUCCESS;
}

extern "C" void* srslte_hip_malloc(size_t nbytes)
{
  void* p = nullptr;
  if (hipMalloc(&p, nbytes ? nbytes : 1) != hipSuccess) {
    hip_log("[srslte_hip] hipMalloc(%zu) failed\n", nbytes);
    return nullptr;
  }
  return p;
}

extern "C" void srslte_hip_free(void* d_ptr)
{
  if (d_ptr) (void)hipFree(d_ptr);
}

extern "C" int srslte_hip_memcpy_h2d(void* d_dst, const void* h_src, size_t nbytes)
{
  HIP_TRY(hipMemcpy(d_dst, h_src, nbytes, hipMemcpyHostToDevice));
  return SRSLTE_SUCCESS;
}

extern "C" int srslte_hip_memcpy_d2h(void* h_dst, const void* d_src, size_t nbytes)
{
  HIP_TRY(hipMemcpy(h_dst, d_src, nbytes, hipMemcpyDeviceToHost));
  return SRSLTE_SUCCESS;
}

extern "C" int srslte_hip_memset(void* d_dst, int value, size_t nbytes)
{
  HIP_TRY(hipMemset(d_dst, value, nbytes));
  return SRSLTE_SUCCESS;
}

extern "C" int srslte_hip_sync(void)
{
  HIP_TRY(hipDeviceSynchronize());
  return SRSLTE_SUCCESS;
}

extern "C" void* srslte_hip_stream_create(void)
{
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
  return (void*)s;
}

extern "C" void srslte_hip_stream_destroy(void* stream)
{
  if (stream) (void)hipStreamDestroy((hipStream_t)stream);
}

extern "C" int srslte_hip_stream_sync(void* stream)
{
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return SRSLTE_SUCCESS;
}

// Timing helper for bench.py: HIP events on the caller's stream (torch.cuda.Event only sees torch's current stream).
extern "C" void* srslte_hip_event_create(void)
{
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return (void*)e;
}
extern "C" int srslte_hip_event_record(void* ev, void* stream)
{
  HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return SRSLTE_SUCCESS;
}
extern "C" float srslte_hip_event_elapsed_ms(void* start, void* stop)
{
  float ms = -1.f;
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.f;
  if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.f;
  return ms;
}
extern "C" void srslte_hip_event_destroy(void* ev)
{
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}
