// Library-level entry points: version and thread-local error string.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void mmsa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mmsa_last_error(void) { return g_err; }
extern "C" int mmsa_version(void) { return 100; }  // 0.1.0

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
