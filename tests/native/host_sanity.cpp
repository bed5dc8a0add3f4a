// host_sanity.cpp -- the host-only parts of the C-ABI layer (weight packer, Powell state machine) built
// with -fsanitize=address,undefined by tests/test_host_sanitizers.py (GPU sanitizers are not available on
// the pool; the device code is exercised by the -m gpu tests).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/nlml_hpe.h"
#include "../../nlml_hpe_amd/csrc/abi_internal.h"
#include "../../nlml_hpe_amd/csrc/layout.h"
#include "../../nlml_hpe_amd/csrc/powell.h"

namespace nlml {
int fail(int code, const char* msg) { std::fprintf(stderr, "fail(%d): %s\n", code, msg); return code ? code : -1; }
}  // namespace nlml

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

static int check_pack(int F, int mode) {
  const int encN[6] = {1024, 512, 256, 128, 64, 9}, encK[6] = {F, 1024, 512, 256, 128, 64};
  const int headN[5] = {128, 256, 128, 64, 1}, headK[5] = {3, 128, 256, 128, 64};
  unsigned seed = 12345u + F;
  std::vector<std::vector<float>> store;
  const float* enc_w[6]; const float* enc_b[6]; const float* head_w[3][5]; const float* head_b[3][5];
  auto mk = [&](size_t n) { store.emplace_back(n); for (auto& v : store.back()) v = frand(seed); return store.back().data(); };
  for (int i = 0; i < 6; ++i) { enc_w[i] = mk((size_t)encN[i] * encK[i]); enc_b[i] = mk(encN[i]); }
  for (int g = 0; g < 3; ++g) for (int i = 0; i < 5; ++i) { head_w[g][i] = mk((size_t)headN[i] * headK[i]); head_b[g][i] = mk(headN[i]); }
  const size_t n = nlml::blob_bytes_for(F, mode);
  std::vector<unsigned char> blob(n);                       // exact size: any overrun is caught by ASan
  if (nlml::pack_blob(F, mode, enc_w, enc_b, head_w, head_b, blob.data(), n) != 0) return 1;
  if (nlml::pack_blob(F, mode, enc_w, enc_b, head_w, head_b, blob.data(), n - 16) == 0) return 2;   // too small must fail
  const nlml::Header* h = reinterpret_cast<const nlml::Header*>(blob.data());
  if (h->magic != nlml::BLOB_MAGIC || h->F != (unsigned)F || (size_t)h->total16 * 16 != n) return 3;
  return 0;
}

static int check_powell() {
  nlml::PowellState s;
  double x0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  nlml::powell_init(s, x0);
  double f = 0.0;
  int evals = 0;
  while (nlml::powell_step(s, f)) {
    f = 0.0;
    for (int k = 0; k < 8; ++k) f += (k + 1) * (s.xeval[k] - 0.25 * k) * (s.xeval[k] - 0.25 * k);
    if (++evals > 100000) return 10;
  }
  for (int k = 0; k < 8; ++k) if (std::fabs(s.x[k] - 0.25 * k) > 1e-3) return 11;
  return s.status == nlml::PW_CONVERGED ? 0 : 12;
}

int main() {
  for (int mode = 0; mode < 2; ++mode)
    for (int F : {1404, 136, 13, 1, 64, 2000})
      if (int rc = check_pack(F, mode)) { std::printf("pack F=%d mode=%d failed: %d\n", F, mode, rc); return 1; }
  if (int rc = check_powell()) { std::printf("powell failed: %d\n", rc); return 1; }
  std::printf("host sanity ok\n");
  return 0;
}
