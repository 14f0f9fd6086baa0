// Library-level entry points: version and thread-local error string.
#include "common.h"
#include "../../include/mmsa_version.h"
#include <string.h>

static thread_local char g_err[512] = "";

void mmsa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mmsa_last_error(void) { return g_err; }
extern "C" int mmsa_version(void) { return MMSA_ABI_VERSION; }   // include/mmsa.h; mmsa/lib.py refuses a library whose number differs from the one it was written for

// HIP-event helpers so bench.py can time kernels on the launch stream without torch.cuda.Event.
extern "C" int mmsa_event_create(void** ev) {
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) { mmsa_set_error("hipEventCreate failed"); return MMSA_ERR_LAUNCH; }
  *ev = (void*)e;
  return MMSA_OK;
}
extern "C" int mmsa_event_record(void* ev, hipStream_t s) {
  if (hipEventRecord((hipEvent_t)ev, s) != hipSuccess) { mmsa_set_error("hipEventRecord failed"); return MMSA_ERR_LAUNCH; }
  return MMSA_OK;
}
extern "C" int mmsa_event_elapsed_ms(void* a, void* b, float* ms) {
  if (hipEventSynchronize((hipEvent_t)b) != hipSuccess) { mmsa_set_error("hipEventSynchronize failed"); return MMSA_ERR_LAUNCH; }
  if (hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) { mmsa_set_error("hipEventElapsedTime failed"); return MMSA_ERR_LAUNCH; }
  return MMSA_OK;
}
extern "C" int mmsa_event_destroy(void* ev) { hipEventDestroy((hipEvent_t)ev); return MMSA_OK; }

// Testing aid (tests/test_backbone_gpu.py, MMSA_DEBUG_POISON_LDS=1 in mmsa/lib.py): overwrite the LDS of every CU with `pattern`
// so that a kernel which reads LDS it has not written itself (the leftovers of whatever ran on that CU before) shows up as NaNs
// or as a changed result.  One 160-KiB workgroup fills a CU's LDS; 4 x #CUs workgroups reach every CU.
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned pattern) {
  extern __shared__ unsigned lds_words[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) lds_words[i] = pattern;
  __syncthreads();
  if (lds_words[(threadIdx.x * 97) % (160 * 1024 / 4)] != pattern) asm volatile("s_nop 0");   // keep the stores
}
extern "C" int mmsa_debug_poison_lds(unsigned pattern, hipStream_t stream) {
  static MmsaPerDevice per_dev_ = {};
  (void)mmsa_per_device(per_dev_, [] { (void)hipFuncSetAttribute((const void*)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), 160 * 1024, stream, pattern);
  MMSA_CHECK_LAUNCH("debug_poison_lds");
  return MMSA_OK;
}
