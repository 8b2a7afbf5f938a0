/* LD_PRELOAD shim (debug aid): logs every hipMalloc / hipFree of the process to stderr, so that the address of a
 * "Memory access fault by GPU" can be placed against the device allocations that were live.  gcc -shared -fPIC -o hip_alloc_log.so hip_alloc_log.c -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
typedef int hipError_t;
hipError_t hipMalloc(void** p, size_t n) {
  static hipError_t (*real)(void**, size_t);
  if (!real) real = (hipError_t(*)(void**, size_t))dlsym(RTLD_NEXT, "hipMalloc");
  hipError_t e = real(p, n);
  fprintf(stderr, "ALLOC %p %p %zu\n", *p, (void*)((char*)*p + n), n);
  return e;
}
hipError_t hipFree(void* p) {
  static hipError_t (*real)(void*);
  if (!real) real = (hipError_t(*)(void*))dlsym(RTLD_NEXT, "hipFree");
  fprintf(stderr, "FREE %p\n", p);
  return real(p);
}
