// host build of the lazy Fp6 product for tests/test_lazy_tower_model.py: g++ -O1 -std=c++17 -fPIC -shared -o liblazy_tower.so lazy_tower_host.cpp
#include "lazy_tower.h"
extern "C" void lz_fp6_mul(const int32_t* a, const int32_t* b, int32_t* out) {
  lz::F6 x, y;
  lz::F* xp[6] = {&x.c0.c0, &x.c0.c1, &x.c1.c0, &x.c1.c1, &x.c2.c0, &x.c2.c1};
  lz::F* yp[6] = {&y.c0.c0, &y.c0.c1, &y.c1.c0, &y.c1.c1, &y.c2.c0, &y.c2.c1};
  for (int c = 0; c < 6; ++c) for (int k = 0; k < 10; ++k) { xp[c]->l[k] = a[10 * c + k]; yp[c]->l[k] = b[10 * c + k]; }
  lz::F6 r = lz::f6_norm(lz::fp6_mul_lazy(x, y));
  lz::F* rp[6] = {&r.c0.c0, &r.c0.c1, &r.c1.c0, &r.c1.c1, &r.c2.c0, &r.c2.c1};
  for (int c = 0; c < 6; ++c) for (int k = 0; k < 10; ++k) out[10 * c + k] = rp[c]->l[k];
}
