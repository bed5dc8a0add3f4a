// Host build of the device's correctly rounded cos (nlml_hpe_amd/csrc/cr_cos.h) for tests/test_abi_and_host.py.
#include "../../nlml_hpe_amd/csrc/cr_cos.h"
extern "C" void cr_cos_array(const double* x, double* y, long n) {
  for (long i = 0; i < n; ++i) y[i] = nlml::cr_cos(x[i]);
}
// the f-vector entry both ways: the fast form (library cos unless the float32 could change) and the slow path alone
extern "C" void cr_fvalue_arrays(const double* a, const double* t, const double* d, float* fast, float* slow, long n) {
  for (long i = 0; i < n; ++i) {
    fast[i] = nlml::cr_f32_a_cos_d(a[i], t[i], d[i]);
    slow[i] = (float)(a[i] * nlml::cr_cos(t[i]) + d[i]);
  }
}
