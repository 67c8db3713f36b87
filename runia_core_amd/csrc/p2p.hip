// (e) multi-GPU: one-shot all-gather of the score shards over xGMI (SURVEY section 5's fallback for a latency-bound
// gather).  The path has exactly one exchange per postprocessor call - every rank's (N / world,) score shard to every
// rank - and at cfg2 a shard is 80 KB: a ring all_gather of that size is 2 (world - 1) latency hops, not
// bandwidth.  Here every rank WRITES its shard straight into every peer's receive buffer (mapped through HIP IPC;
// xGMI is point to point, so the world - 1 writes go out on world - 1 different links at once) and raises a flag beside
// it; further workgroups of the SAME launch wait for the world flags of this step and copy the gathered vector out.  One
// launch, no host synchronisation, no ring.
//
// Buffer of a rank (fine-grained device memory, so that a peer's writes are visible without a cache flush):
//   [ flags: 2 slots x 64 ranks x u64 | status u32 | per-peer completion counters | pad to 4 KiB | slot 0 | slot 1 ]
// each slot holding world x shard_capacity bytes.  Step `seq` (1, 2, 3, ...) uses slot seq & 1 and flag value seq:
// a slot is rewritten two steps later, by which time every peer has passed the wait of the step in between, which its
// stream orders after its own readers of the older slot (the copy-out below).
//
// Every wave that spins has an exit: the wait gives up after `timeout_ms` of the constant-rate clock, sets the status
// word and still returns (the caller reads the status with runia_p2p_status; it never hangs the GPU).
#include "common.hpp"

#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {

constexpr int kMaxRanks = 64;
constexpr size_t kHeader = 4096;
constexpr size_t kFlagsBytes = 2 * kMaxRanks * sizeof(unsigned long long);  // 1 024
constexpr size_t kStatusOff = kFlagsBytes;                                    // u32 status (0 ok, 1 timed out)
constexpr size_t kDoneOff = kFlagsBytes + 64;                                 // u32 [2][kMaxRanks] block counters
constexpr size_t kCopyDoneOff = kDoneOff + 2 * kMaxRanks * sizeof(unsigned);  // u32 [2][kMaxRanks] copy-out part counters
constexpr size_t kAckOff = kCopyDoneOff + 2 * kMaxRanks * sizeof(unsigned);   // u64 [2][kMaxRanks] last step copied out of (slot, source)
static_assert(kAckOff + 2 * kMaxRanks * sizeof(unsigned long long) <= kHeader, "header layout");
// Slot reuse rests on an ordering argument (top of the file): rank A overwrites peer B's slot at step s only after A's own
// step s - 1 launch, which waited for B's flag s - 1, which B's stream raised after B's step s - 2 copy-out.  With the
// check switched on (runia_p2p_debug) the argument is ASSERTED: every copy-out records the step it finished in the
// reader's buffer (ack[slot][source]), and a writer reads the peer's ack over the link before it touches the slot - it has
// to be exactly s - 2, else status bit 1 (value 2) is set.  The acks are always written (local); the switch gates the remote
// read only, so it may be flipped at any step and need not flip on every rank at once.  Costs one remote read per push; off
// by default.
std::atomic<int> g_debug{0};

struct PeerTable { char* buf[16]; };  // by value in the kernel arguments (world <= 16)

std::mutex g_mu;
std::unordered_map<void*, size_t> g_owned;   // buffers of this process -> bytes
std::unordered_map<void*, int> g_opened;      // peer mappings of this process

__device__ __forceinline__ void p2p_push(const char* __restrict__ shard, size_t shard_bytes, size_t cap,
                                         const PeerTable& peers, char* self, int world, int rank, unsigned long long seq,
                                         int blocks_per_peer, int p, int j, int debug) {
  const int slot = (int)(seq & 1ull);
  if (debug && j == 0 && threadIdx.x == 0 && seq > 2ull) {  // the peer has copied my step seq - 2 out of this slot
    const unsigned long long* ack = reinterpret_cast<const unsigned long long*>(peers.buf[p] + kAckOff) + slot * kMaxRanks + rank;
    if (__hip_atomic_load(ack, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq - 2ull)
      atomicOr(reinterpret_cast<unsigned*>(self + kStatusOff), 2u);
  }
  char* dst = peers.buf[p] + kHeader + (size_t)slot * (size_t)world * cap + (size_t)rank * cap;
  const size_t chunk = ((shard_bytes + blocks_per_peer - 1) / blocks_per_peer + 15) & ~(size_t)15;
  const size_t lo = (size_t)j * chunk, hi = (lo + chunk < shard_bytes) ? lo + chunk : shard_bytes;
  if (lo < hi) {
    if (((((uintptr_t)shard) | ((uintptr_t)dst)) & 15) == 0) {
      const size_t n16 = (hi - lo) / 16;
      const uint4* s4 = reinterpret_cast<const uint4*>(shard + lo);
      uint4* d4p = reinterpret_cast<uint4*>(dst + lo);
      for (size_t i = threadIdx.x; i < n16; i += 256) d4p[i] = s4[i];
      for (size_t i = lo + n16 * 16 + threadIdx.x; i < hi; i += 256) dst[i] = shard[i];
    } else {
      for (size_t i = lo + threadIdx.x; i < hi; i += 256) dst[i] = shard[i];
    }
  }
  __threadfence_system();  // this block's writes are out before its count is
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* done = reinterpret_cast<unsigned*>(self + kDoneOff) + slot * kMaxRanks + p;
    const unsigned before = atomicAdd(done, 1u);
    if (before + 1u == (unsigned)blocks_per_peer) {  // the last block of this peer's copy raises the flag
      atomicExch(done, 0u);
      __threadfence_system();
      unsigned long long* flag = reinterpret_cast<unsigned long long*>(peers.buf[p]) + slot * kMaxRanks + rank;
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__device__ __forceinline__ void p2p_wait_copy(char* self, char* __restrict__ out, size_t shard_bytes, size_t cap,
                                              int world, unsigned long long seq, long long timeout_ticks, int r, int part,
                                              int parts, int debug) {
  const int slot = (int)(seq & 1ull);
  __shared__ int ok;
  if (threadIdx.x == 0) {
    const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(self) + slot * kMaxRanks + r;
    const long long t0 = wall_clock64();
    int good = 1;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > timeout_ticks) {  // a peer never arrived: report, do not hang
        good = 0;
        atomicOr(reinterpret_cast<unsigned*>(self + kStatusOff), 1u);
        break;
      }
    }
    ok = good;
  }
  __syncthreads();
  __threadfence_system();
  const char* src = self + kHeader + (size_t)slot * (size_t)world * cap + (size_t)r * cap;
  char* dst = out + (size_t)r * shard_bytes;
  // this block's part of the shard (reads of fine-grained memory are not cached: many short copies in parallel)
  const size_t chunk = ((shard_bytes + parts - 1) / parts + 15) & ~(size_t)15;
  const size_t lo = (size_t)part * chunk, hi = (lo + chunk < shard_bytes) ? lo + chunk : shard_bytes;
  if (lo < hi) {
    if (((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
      const size_t n16 = (hi - lo) / 16;
      const uint4* s4 = reinterpret_cast<const uint4*>(src + lo);
      uint4* d4p = reinterpret_cast<uint4*>(dst + lo);
      for (size_t i = threadIdx.x; i < n16; i += 256) d4p[i] = s4[i];
      for (size_t i = lo + n16 * 16 + threadIdx.x; i < hi; i += 256) dst[i] = src[i];
    } else {
      for (size_t i = lo + threadIdx.x; i < hi; i += 256) dst[i] = src[i];
    }
  }
  (void)ok;
  (void)debug;
  {  // The last part of this shard's copy-out records the step (see g_debug).  ALWAYS recorded (local atomics only): the
     // assertion can then be switched on at any step of a live buffer, and by the ranks at different steps, without a false
     // alarm (acks used to be written only by launches that ran with the check on) - only the remote READ is gated.
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned* cd = reinterpret_cast<unsigned*>(self + kCopyDoneOff) + slot * kMaxRanks + r;
      if (atomicAdd(cd, 1u) + 1u == (unsigned)parts) {
        atomicExch(cd, 0u);
        unsigned long long* ack = reinterpret_cast<unsigned long long*>(self + kAckOff) + slot * kMaxRanks + r;
        __hip_atomic_store(ack, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ONE launch: blocks [0, world * bpp) push this rank's shard to the peers, blocks [world * bpp, world * bpp + world) wait
// for the flag of one source rank each and copy its shard out.  The waiting blocks depend on pushes of OTHER processes
// (and, for the own shard, on the push blocks of this grid, which have lower ids and are dispatched first; the whole
// grid - at most 2 * 16 * 32 workgroups of 256 threads - is resident at once, so a waiting block never keeps a push block off the chip).
__global__ __launch_bounds__(256) void p2p_gather_kernel(const char* __restrict__ shard, size_t shard_bytes, size_t cap,
                                                         PeerTable peers, char* self, char* __restrict__ out, int world,
                                                         int rank, unsigned long long seq, int blocks_per_peer,
                                                         long long timeout_ticks, int debug) {
  const int push_blocks = world * blocks_per_peer;
  if ((int)blockIdx.x < push_blocks) {
    p2p_push(shard, shard_bytes, cap, peers, self, world, rank, seq, blocks_per_peer, (int)blockIdx.x / blocks_per_peer,
             (int)blockIdx.x % blocks_per_peer, debug);
  } else {
    const int w = (int)blockIdx.x - push_blocks;  // the copy-out of a shard is cut like its push
    p2p_wait_copy(self, out, shard_bytes, cap, world, seq, timeout_ticks, w / blocks_per_peer, w % blocks_per_peer,
                  blocks_per_peer, debug);
  }
}

}  // namespace

extern "C" size_t runia_p2p_buffer_bytes(int world, size_t shard_capacity_bytes) {
  if (world < 1 || world > 16 || shard_capacity_bytes == 0) return 0;
  const size_t cap = (shard_capacity_bytes + 255) & ~(size_t)255;
  return kHeader + 2 * (size_t)world * cap;
}

// Allocate (and zero) this rank's receive buffer: fine-grained device memory of runia_p2p_buffer_bytes bytes.
extern "C" int runia_p2p_alloc(int world, size_t shard_capacity_bytes, void** buffer) {
  const size_t bytes = runia_p2p_buffer_bytes(world, shard_capacity_bytes);
  if (!bytes || !buffer) return RUNIA_E_INVALID;
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess || !p) return RUNIA_E_LAUNCH;
  if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(p);
    return RUNIA_E_LAUNCH;
  }
  std::lock_guard<std::mutex> lock(g_mu);
  g_owned[p] = bytes;
  *buffer = p;
  return RUNIA_OK;
}

extern "C" int runia_p2p_free(void* buffer) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_owned.find(buffer);
  if (it == g_owned.end()) return RUNIA_E_INVALID;
  g_owned.erase(it);
  return hipFree(buffer) == hipSuccess ? RUNIA_OK : RUNIA_E_LAUNCH;
}

// 64-byte HIP IPC handle of a buffer from runia_p2p_alloc (to be sent to the peers by any host channel).
extern "C" int runia_p2p_export(void* buffer, void* handle64) {
  if (!handle64) return RUNIA_E_INVALID;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_owned.find(buffer) == g_owned.end()) return RUNIA_E_INVALID;
  }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handles are 64 bytes");
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, buffer) != hipSuccess) return RUNIA_E_LAUNCH;
  std::memcpy(handle64, reinterpret_cast<const void*>(&h), 64);
  return RUNIA_OK;
}

extern "C" int runia_p2p_open(const void* handle64, void** peer_buffer) {
  if (!handle64 || !peer_buffer) return RUNIA_E_INVALID;
  hipIpcMemHandle_t h;
  std::memcpy(reinterpret_cast<void*>(&h), handle64, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p) return RUNIA_E_LAUNCH;
  std::lock_guard<std::mutex> lock(g_mu);
  g_opened[p] = 1;
  *peer_buffer = p;
  return RUNIA_OK;
}

extern "C" int runia_p2p_close(void* peer_buffer) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_opened.find(peer_buffer);
  if (it == g_opened.end()) return RUNIA_E_INVALID;
  g_opened.erase(it);
  return hipIpcCloseMemHandle(peer_buffer) == hipSuccess ? RUNIA_OK : RUNIA_E_LAUNCH;
}

// One-shot all-gather, stream-ordered: `local_shard` (shard_bytes, device) is written into slot seq & 1 of every rank's
// buffer (peer_buffers[r] = rank r's buffer as mapped in THIS process; peer_buffers[rank] = this rank's own), then `out`
// (world x shard_bytes, device) receives the gathered vector once all `world` flags of step `seq` are up.  seq = 1, 2, 3,
// ... must advance by one per call on every rank; consecutive calls alternate slots (see the header of this file).
extern "C" int runia_p2p_all_gather(const void* local_shard, size_t shard_bytes, void* out, void* const* peer_buffers,
                                    int world, int rank, size_t shard_capacity_bytes, uint64_t seq, int timeout_ms,
                                    runia_stream_t stream) {
  if (world < 1 || world > 16 || rank < 0 || rank >= world || !local_shard || !out || !peer_buffers || seq == 0)
    return RUNIA_E_INVALID;
  if (shard_bytes == 0 || shard_bytes > shard_capacity_bytes || timeout_ms < 1) return RUNIA_E_INVALID;
  const size_t cap = (shard_capacity_bytes + 255) & ~(size_t)255;
  PeerTable t{};
  {
    std::lock_guard<std::mutex> lock(g_mu);
    for (int r = 0; r < world; ++r) {
      void* b = peer_buffers[r];
      if (!b) return RUNIA_E_INVALID;
      if (r == rank) {
        auto it = g_owned.find(b);
        if (it == g_owned.end() || it->second < kHeader + 2 * (size_t)world * cap) return RUNIA_E_WORKSPACE;
      } else if (g_opened.find(b) == g_opened.end()) {
        return RUNIA_E_INVALID;  // not a mapping made by runia_p2p_open in this process
      }
      t.buf[r] = reinterpret_cast<char*>(b);
    }
  }
  char* self = t.buf[rank];
  int bpp = (int)((shard_bytes + 8191) / 8192);  // <= 8 KiB per workgroup: two 16-byte passes of 256 threads
  bpp = bpp < 1 ? 1 : (bpp > 32 ? 32 : bpp);
  hipStream_t s = as_stream(stream);
  const long long ticks = (long long)timeout_ms * 100000ll;  // wall_clock64 runs at 100 MHz
  p2p_gather_kernel<<<(unsigned)(2 * world * bpp), 256, 0, s>>>(reinterpret_cast<const char*>(local_shard), shard_bytes,
                                                                     cap, t, self, reinterpret_cast<char*>(out), world, rank,
                                                                     (unsigned long long)seq, bpp, ticks,
                                                                     g_debug.load(std::memory_order_relaxed));
  return runia_check_launch();
}

// Switch the slot-reuse assertion on / off for the launches that follow (every rank of a group alike: the writers read
// what the readers record).  Returns the previous setting.
extern "C" int runia_p2p_debug(int on) { return g_debug.exchange(on ? 1 : 0); }

// Bit 0 (1): a wait timed out (a peer never arrived); bit 1 (2): the slot-reuse assertion of runia_p2p_debug failed.
// 0 = neither so far.  Synchronises the device.
extern "C" int runia_p2p_status(void* buffer, int* status) {
  if (!status) return RUNIA_E_INVALID;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_owned.find(buffer) == g_owned.end()) return RUNIA_E_INVALID;
  }
  unsigned v = 0;
  if (hipMemcpy(&v, reinterpret_cast<char*>(buffer) + kStatusOff, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
    return RUNIA_E_LAUNCH;
  *status = (int)v;
  return RUNIA_OK;
}
