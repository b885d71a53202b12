// KV-block transport between a prompt worker and its token worker (Splitwise hand-off).
//
// What it replaces: the reference's three one-thread kernels that enqueue put / signal / wait
// requests into an MSCCL++ proxy FIFO served by a CPU thread (csrc/kv_comm_kernels.cu:3-29,
// vllm/worker/comm_utils.py:90-147).  Here the GPU moves the bytes itself: the token
// worker's cache is mapped into the prompt worker's address space (HIP IPC over xGMI) and
//   * sw_kv_push   copies byte ranges local cache -> peer cache, same offset on both sides
//                  (comm_utils.py:132-147), 16 bytes per lane, any number of ranges per launch;
//   * sw_kv_push_mapped  the same with a destination offset of its own per range: a prompt worker that
//                  keeps only the prompts in flight, under block numbers of its own;
//   * sw_kv_signal publishes a per-sequence hand-off flag in the peer's memory after a
//                  system-scope release (ordered after every earlier push on the stream);
//   * sw_kv_wait   spins (bounded, like the reference's 1e8 bound) on the local flag with
//                  system-scope loads, then acquires, so later kernels read the pushed KV.
// Flags carry an epoch (hand-off count of the slot), never reset, so a slot is reusable
// without a clearing step.
#include "common.cuh"

#include <algorithm>
#include <string.h>

namespace sw {

constexpr int kMaxRuns = 48;
constexpr int kPushThreads = 256;
constexpr int64_t kChunkBytes = 64 * 1024;          // bytes one workgroup moves

struct PushRuns {
  int64_t offset[kMaxRuns];     // byte offset from the destination base, multiple of 16
  int64_t src_offset[kMaxRuns]; // byte offset from the source base, multiple of 16
  int64_t nbytes[kMaxRuns];     // multiple of 16
  int32_t first_chunk[kMaxRuns + 1];
  int32_t num_runs;
};

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// Stores go out write-through at system scope (sc0 sc1): the bytes leave this GPU's caches as
// they are issued, so no cache write-back is needed before the hand-off flag is raised; each
// wave only drains its own store queue before it retires.
// A workgroup walks the chunks blockIdx.x, blockIdx.x + gridDim.x, ...: the grid may be capped to the
// few workgroups one xGMI link can keep busy, so that a push running beside a layer's attention (on
// its own stream) does not park thousands of workgroups on link-bound stores.  With few workgroups every one
// of them has to keep the link busy by itself: a thread holds a whole chunk's 16 pieces in registers, and the
// loads of the next chunk are issued BEFORE the stores of this one -- the memory counter retires in issue
// order, so waiting for those loads then does not wait for this chunk's stores, which get a whole chunk's
// time to cross the link.
constexpr int kPushPieces = (int)(kChunkBytes / 16 / kPushThreads);     // 16-byte pieces per thread and chunk

__global__ void __launch_bounds__(kPushThreads)
kv_push_kernel(char* __restrict__ dst_base, const char* __restrict__ src_base, PushRuns runs, int num_chunks) {
  constexpr int kAuxSystem = 17;                      // sc0 | sc1
  int chunk = blockIdx.x;
  if (chunk >= num_chunks) return;                    // workgroup-uniform
  int r = 0;
  // wave-uniform descriptors of a chunk's source and destination ranges: pieces past the end of a short chunk
  // read as zero and are not stored (bounds-checked by the hardware)
  auto describe = [&](int c, auto& s_rsrc, auto& d_rsrc) {
    while (r + 1 < runs.num_runs && c >= runs.first_chunk[r + 1]) ++r;
    const int64_t begin = (int64_t)(c - runs.first_chunk[r]) * kChunkBytes;
    const int nbytes = (int)(min(begin + kChunkBytes, runs.nbytes[r]) - begin);
    const uint64_t sa = reinterpret_cast<uint64_t>(src_base + runs.src_offset[r] + begin);
    const uint64_t da = reinterpret_cast<uint64_t>(dst_base + runs.offset[r] + begin);
    // (the intrinsic returns a signed int: each half goes through an unsigned 32-bit value, or a low half with its
    // top bit set would sign-extend over the high half)
    const uint32_t s_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sa);
    const uint32_t s_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sa >> 32));
    const uint32_t d_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)da);
    const uint32_t d_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(da >> 32));
    const uint64_t su = ((uint64_t)s_hi << 32) | (uint64_t)s_lo;
    const uint64_t du = ((uint64_t)d_hi << 32) | (uint64_t)d_lo;
    s_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(su), 0, nbytes, 0x00020000);
    d_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(du), 0, nbytes, 0x00020000);
  };
  using rsrc_t = decltype(__builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(nullptr), 0, 0, 0));
  rsrc_t s_cur, d_cur, s_next, d_next;
  u32x4 cur[kPushPieces], next[kPushPieces];
  const int lane_off = threadIdx.x * 16;
  describe(chunk, s_cur, d_cur);
#pragma unroll
  for (int i = 0; i < kPushPieces; ++i)
    cur[i] = __builtin_amdgcn_raw_buffer_load_b128(s_cur, lane_off + i * kPushThreads * 16, 0, 0);
  // (the last chunk's stores are code of their own: at a point that both "more loads were issued" and "none were"
  // reach, the compiler's wait-count bookkeeping has to assume the worse of the two and would make every store
  // wait for the previous chunk's stores)
  for (;;) {
    const int following = chunk + gridDim.x;
    if (following >= num_chunks) break;               // workgroup-uniform
    describe(following, s_next, d_next);
#pragma unroll
    for (int i = 0; i < kPushPieces; ++i)
      next[i] = __builtin_amdgcn_raw_buffer_load_b128(s_next, lane_off + i * kPushThreads * 16, 0, 0);
#pragma unroll
    for (int i = 0; i < kPushPieces; ++i)
      __builtin_amdgcn_raw_buffer_store_b128(cur[i], d_cur, lane_off + i * kPushThreads * 16, 0, kAuxSystem);
#pragma unroll
    for (int i = 0; i < kPushPieces; ++i) cur[i] = next[i];
    d_cur = d_next;
    chunk = following;
  }
#pragma unroll
  for (int i = 0; i < kPushPieces; ++i)
    __builtin_amdgcn_raw_buffer_store_b128(cur[i], d_cur, lane_off + i * kPushThreads * 16, 0, kAuxSystem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void kv_signal_kernel(int32_t* flags, int slot, int32_t epoch) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(flags + slot, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// status[0] is set to 1 + slot when the bound runs out (the caller checks it on the host).
__global__ void kv_wait_kernel(const int32_t* flags, int slot, int32_t epoch, long long max_spins,
                               int32_t* status) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    long long spins = 0;
    bool ok = false;
    while (spins < max_spins) {
      const int32_t v = __hip_atomic_load(flags + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (v - epoch >= 0) { ok = true; break; }
      __builtin_amdgcn_s_sleep(32);
      ++spins;
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);         // system scope
    if (!ok) __hip_atomic_store(status, 1 + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace sw

extern "C" {

int sw_ipc_get_mem_handle(const void* dev_ptr, void* handle_out64) {
  using namespace sw;
  SW_CHECK_ARG(dev_ptr && handle_out64, "ipc_get_mem_handle: NULL argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
  hipIpcMemHandle_t h;
  SW_CHECK_HIP(hipIpcGetMemHandle(&h, const_cast<void*>(dev_ptr)));
  memcpy(handle_out64, &h, sizeof(h));
  return SW_OK;
}

int sw_mem_address_range(const void* dev_ptr, void** base_out, int64_t* size_out) {
  using namespace sw;
  SW_CHECK_ARG(dev_ptr && base_out && size_out, "mem_address_range: NULL argument");
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  SW_CHECK_HIP(hipMemGetAddressRange(&base, &size, const_cast<void*>(dev_ptr)));
  *base_out = base;
  *size_out = (int64_t)size;
  return SW_OK;
}

int sw_ipc_open_mem_handle(const void* handle64, void** dev_ptr_out) {
  using namespace sw;
  SW_CHECK_ARG(handle64 && dev_ptr_out, "ipc_open_mem_handle: NULL argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  SW_CHECK_HIP(hipIpcOpenMemHandle(dev_ptr_out, h, hipIpcMemLazyEnablePeerAccess));
  return SW_OK;
}

int sw_ipc_close_mem_handle(void* dev_ptr) {
  using namespace sw;
  SW_CHECK_HIP(hipIpcCloseMemHandle(dev_ptr));
  return SW_OK;
}

int sw_kv_push_mapped(void* dst_base, const void* src_base, const int64_t* dst_offsets,
                      const int64_t* src_offsets, const int64_t* nbytes, int num_runs, int max_workgroups,
                      void* stream) {
  using namespace sw;
  if (num_runs == 0) return SW_OK;
  SW_CHECK_ARG(dst_base && src_base && dst_offsets && src_offsets && nbytes && num_runs > 0, "kv_push: bad arguments");
  SW_CHECK_ARG(reinterpret_cast<uintptr_t>(dst_base) % 16 == 0 && reinterpret_cast<uintptr_t>(src_base) % 16 == 0,
               "kv_push: bases must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int done = 0;
  while (done < num_runs) {
    PushRuns runs;
    int chunks = 0, n = 0;
    for (; n < kMaxRuns && done + n < num_runs; ++n) {
      const int64_t off = dst_offsets[done + n], soff = src_offsets[done + n], nb = nbytes[done + n];
      SW_CHECK_ARG(off >= 0 && soff >= 0 && nb > 0 && off % 16 == 0 && soff % 16 == 0 && nb % 16 == 0,
                   "kv_push: run %d (offsets %lld <- %lld, %lld bytes) is not 16-byte granular", done + n,
                   (long long)off, (long long)soff, (long long)nb);
      runs.offset[n] = off;
      runs.src_offset[n] = soff;
      runs.nbytes[n] = nb;
      runs.first_chunk[n] = chunks;
      const int64_t c = (nb + kChunkBytes - 1) / kChunkBytes;
      SW_CHECK_ARG(chunks + c < (1 << 30), "kv_push: too many bytes in one call");
      chunks += (int)c;
    }
    runs.first_chunk[n] = chunks;
    runs.num_runs = n;
    const int grid = max_workgroups > 0 ? std::min(chunks, max_workgroups) : chunks;
    kv_push_kernel<<<grid, kPushThreads, 0, s>>>(static_cast<char*>(dst_base),
                                                 static_cast<const char*>(src_base), runs, chunks);
    SW_CHECK_LAUNCH();
    done += n;
  }
  return SW_OK;
}

int sw_kv_push(void* dst_base, const void* src_base, const int64_t* offsets, const int64_t* nbytes,
               int num_runs, int max_workgroups, void* stream) {
  return sw_kv_push_mapped(dst_base, src_base, offsets, offsets, nbytes, num_runs, max_workgroups, stream);
}

int sw_kv_signal(int32_t* flags, int slot, int32_t epoch, void* stream) {
  using namespace sw;
  SW_CHECK_ARG(flags && slot >= 0, "kv_signal: bad arguments");
  kv_signal_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(flags, slot, epoch);
  SW_CHECK_LAUNCH();
  return SW_OK;
}

int sw_kv_wait(const int32_t* flags, int slot, int32_t epoch, int64_t max_spins, int32_t* status,
               void* stream) {
  using namespace sw;
  SW_CHECK_ARG(flags && status && slot >= 0 && max_spins > 0, "kv_wait: bad arguments");
  kv_wait_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(flags, slot, epoch, (long long)max_spins, status);
  SW_CHECK_LAUNCH();
  return SW_OK;
}

}  // extern "C"
