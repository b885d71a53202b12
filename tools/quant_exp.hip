// Tuning harness for the int4 GEMMs (not part of the shipped library): compiles quant_gemm.hip with its
// experiment knobs switched on, times back-to-back launches with HIP events and compares every variant with the
// first one run on the same inputs.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSW_QUANT_EXPERIMENT -I include -I vllm-splitwise_amd/csrc \
//         tools/quant_exp.hip -o gpurun_out/quant_exp
//   SW_QUANT_WAVES=4 SW_QUANT_TMT=4 gpurun_out/quant_exp 256 13824 5120
#include "../vllm-splitwise_amd/csrc/quant_gemm.hip"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

namespace sw {
void set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n");
}
}  // namespace sw

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } \
  } while (0)

struct Variant { const char* name; const char* waves; const char* tmt; const char* chunk; const char* ks; const char* t128 = "0"; };

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 13824, K = argc > 3 ? atoi(argv[3]) : 5120;
  const int group = 128, groups = K / group, reps = 40;
  std::mt19937 rng(1);
  std::vector<uint16_t> a((size_t)M * K), sc((size_t)groups * N);
  std::vector<uint32_t> qw((size_t)(K / 8) * N), qz((size_t)groups * (N / 8));
  std::normal_distribution<float> nd(0.f, 1.f);
  auto f2h = [](float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; };
  for (auto& v : a) v = f2h(nd(rng));
  for (auto& v : sc) v = f2h(1e-4f + 2e-3f * (rng() % 1000) / 1000.f);
  for (auto& v : qw) v = rng();
  for (auto& v : qz) v = rng();
  uint16_t *da, *dsc, *dc; uint32_t *dqw, *dqz; void* ws;
  const int64_t wsb = sw_quant_gemm_workspace_bytes(M, N, K);
  CK(hipMalloc(&da, a.size() * 2)); CK(hipMalloc(&dsc, sc.size() * 2)); CK(hipMalloc(&dc, (size_t)M * N * 2));
  CK(hipMalloc(&dqw, qw.size() * 4)); CK(hipMalloc(&dqz, qz.size() * 4)); CK(hipMalloc(&ws, wsb));
  CK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsc, sc.data(), sc.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dqw, qw.data(), qw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dqz, qz.data(), qz.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(ws, 0, wsb));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<Variant> vars = {
      {"w8 mt2 c256 (shipped)", "8", "2", "256", nullptr},
      {"t128 kq2 auto", "8", "2", "256", nullptr, "2"},
      {"t128 kq2 ks1", "8", "2", "256", "1", "2"},
      {"t128 kq2 ks2", "8", "2", "256", "2", "2"},
      {"t128 kq2 ks3", "8", "2", "256", "3", "2"},
      {"t128 kq2 ks4", "8", "2", "256", "4", "2"},
      {"t128 kq4 auto", "8", "2", "256", nullptr, "4"},
      {"t128 kq4 ks1", "8", "2", "256", "1", "4"},
      {"t128 kq8 auto", "8", "2", "256", nullptr, "8"},
      {"t128 kq8 ks1", "8", "2", "256", "1", "8"},
      {"t64 kq8 auto", "8", "2", "256", nullptr, "68"},
      {"t64 kq8 ks1", "8", "2", "256", "1", "68"},
      {"t32 kq8 auto", "8", "2", "256", nullptr, "38"},
      {"t32 kq8 ks1", "8", "2", "256", "1", "38"},
      {"t32 kq8 ks2", "8", "2", "256", "2", "38"},
      {"t256 kq2 auto", "8", "2", "256", nullptr, "82"},
      {"t256 kq2 ks1", "8", "2", "256", "1", "82"},
      {"t256 kq4cg1 ks1", "8", "2", "256", "1", "84"},
      {"t256 kq4cg1 ks2", "8", "2", "256", "2", "84"},
      {"t256 kq2 ks2", "8", "2", "256", "2", "82"},
      {"t256 kq2 ks3", "8", "2", "256", "3", "82"},
      {"t256 kq2 ks4", "8", "2", "256", "4", "82"},
      {"t128 kq4 ks2", "8", "2", "256", "2", "4"},
      {"t128 kq4 ks3", "8", "2", "256", "3", "4"},
      {"t128 kq4 ks4", "8", "2", "256", "4", "4"},
      {"w8 mt2 ks1", "8", "2", "256", "1"},
      {"shipped path ks2", "8", "2", "256", "2"},
      {"shipped path ks3", "8", "2", "256", "3"},
      {"shipped path ks4", "8", "2", "256", "4"},
  };
  std::vector<uint16_t> ref((size_t)M * N), got((size_t)M * N);
  const char* only = argc > 4 ? argv[4] : nullptr;
  for (size_t v = 0; v < vars.size(); ++v) {
    if (only && v != 0 && !strstr(vars[v].name, only)) continue;
    setenv("SW_QUANT_WAVES", vars[v].waves, 1); setenv("SW_QUANT_TMT", vars[v].tmt, 1);
    setenv("SW_QUANT_CHUNK", vars[v].chunk, 1);
    setenv("SW_QUANT_T128", vars[v].t128, 1);
    if (vars[v].ks) setenv("SW_QUANT_KSPLIT", vars[v].ks, 1); else unsetenv("SW_QUANT_KSPLIT");
    CK(hipMemsetAsync(dc, 0, (size_t)M * N * 2, st));
    CK(hipMemsetAsync(ws, 0, 16384, st));
    auto run = [&]() { return sw_gptq_gemm(dc, da, dqw, dqz, dsc, nullptr, M, N, K, groups, 1, 4, ws, wsb, st); };
    if (run() != 0) { printf("%-24s refused\n", vars[v].name); continue; }
    CK(hipStreamSynchronize(st));
    for (int i = 0; i < 3; ++i) run();
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(got.data(), dc, got.size() * 2, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    if (v == 0) ref = got;
    for (size_t i = 0; i < got.size(); ++i) {
      _Float16 x, y; memcpy(&x, &got[i], 2); memcpy(&y, &ref[i], 2);
      maxerr = std::max(maxerr, (double)std::fabs((float)x - (float)y));
      maxref = std::max(maxref, (double)std::fabs((float)y));
    }
    const double us = ms * 1e3 / reps;
    printf("%-24s %7.1f us %6.1f TFLOP/s   max|diff| %.4g (max|ref| %.3g)\n", vars[v].name, us,
           2.0 * M * N * K / us / 1e6, maxerr, maxref);
    fflush(stdout);
  }
  return 0;
}
