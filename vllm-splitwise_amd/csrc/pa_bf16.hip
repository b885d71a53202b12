// Paged-attention instantiations for activations=bf16_t, fp8-e5m2 cache=false.
// Split per (activation, cache) pair so the translation units build in parallel.
#include "paged_attention.cuh"

namespace sw {
int pa_dispatch_bf16(const PaArgs& a) { return dispatch_pa<bf16_t, false, 8>(a); }
}  // namespace sw
