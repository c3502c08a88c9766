/* A plain-C consumer of include/blsbn254.h: what a cgo / Rust-FFI / C caller sees.  Built and run by tests/test_abi.py with
 * gcc -std=c99 -Wall -Wextra -Werror: the header must be valid C (not only C++), and the shared library must resolve every
 * symbol it uses without any HIP header on the include path.  On a box without a gfx950 device ctx_create reports
 * BLSBN254_E_NO_DEVICE (there is no CPU fallback); with one it creates and destroys a context. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "blsbn254.h"

typedef const char* (*strerror_fn)(int);
typedef int (*create_fn)(int, blsbn254_ctx**);
typedef void (*destroy_fn)(blsbn254_ctx*);
typedef int (*verify_fn)(blsbn254_ctx*, const uint8_t*, const uint8_t*, const uint64_t*, const uint8_t*, size_t, const uint8_t*, size_t, uint8_t*);

int main(int argc, char** argv) {
  void* h;
  strerror_fn se; create_fn cc; destroy_fn cd; verify_fn vb;
  blsbn254_ctx* c = NULL;
  int rc;
  if (argc < 2) return 2;
  h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "%s\n", dlerror()); return 3; }
  *(void**)(&se) = dlsym(h, "blsbn254_strerror");
  *(void**)(&cc) = dlsym(h, "blsbn254_ctx_create");
  *(void**)(&cd) = dlsym(h, "blsbn254_ctx_destroy");
  *(void**)(&vb) = dlsym(h, "blsbn254_verify_batch");
  if (!se || !cc || !cd || !vb) return 4;
  rc = cc(0, &c);
  if (rc == 0) {
    uint64_t off[1] = {0};
    if (vb(c, NULL, NULL, off, NULL, 0, NULL, 0, NULL) != 0) return 5;      /* empty batch: ok, nothing written */
    cd(c);
    printf("ctx ok\n");
  } else {
    if (rc != BLSBN254_E_NO_DEVICE || c != NULL) return 6;
    printf("no device: %s\n", se(rc));
  }
  if (strcmp(se(2), "invalid G1 bytes") != 0) return 7;
  return 0;
}
