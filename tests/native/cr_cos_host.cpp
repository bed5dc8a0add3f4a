// Host build of the device's correctly rounded cos (nlml_hpe_amd/csrc/cr_cos.h) for tests/test_abi_and_host.py.
#include "../../nlml_hpe_amd/csrc/cr_cos.h"
extern "C" void cr_cos_array(const double* x, double* y, long n) {
  for (long i = 0; i < n; ++i) y[i] = nlml::cr_cos(x[i]);
}
