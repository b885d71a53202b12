// Paged-attention instantiations for activations=float, fp8-e5m2 cache=true.
// Split per (activation, cache) pair so the translation units build in parallel.
#include "paged_attention.cuh"

namespace sw {
int pa_dispatch_f32_fp8(const PaArgs& a) { return dispatch_pa<float, true, 1>(a); }
}  // namespace sw
