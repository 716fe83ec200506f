// Test tool (not part of the product): LD_PRELOAD interposer that fills every hipMalloc'ed block with a poison byte, so that a kernel
// reading device memory nobody wrote shows up deterministically instead of depending on what the allocator hands back.
//   g++ -shared -fPIC -O1 tests/tools/poison_hipmalloc.cpp -o /tmp/libpoison.so -ldl -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -L/opt/rocm/lib -lamdhip64
//   LD_PRELOAD=/tmp/libpoison.so POISON_BYTE=165 python -m pytest tests -q -m gpu
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdlib.h>

extern "C" hipError_t hipMalloc(void** ptr, size_t size)
{
  typedef hipError_t (*fn_t)(void**, size_t);
  static fn_t real = (fn_t)dlsym(RTLD_NEXT, "hipMalloc");
  const hipError_t e = real(ptr, size);
  if (e == hipSuccess && ptr && *ptr && size) {
    const char* b = getenv("POISON_BYTE");
    (void)hipMemset(*ptr, b ? atoi(b) : 0xA5, size);
    (void)hipDeviceSynchronize();
  }
  return e;
}
