// Guard-page device allocator for memory-safety runs of the HIP path ("ASan-instrumented HIP builds / run-to-run
// checks"; VERDICT r03 "what's missing" item 1).  TEST INFRASTRUCTURE - never loaded by the product.
//
// Plugged into PyTorch with torch.cuda.memory.CUDAPluggableAllocator(lib, "efence_malloc", "efence_free") (tools/efence/efence.py).
// Every tensor gets its OWN virtual-address range built with the HIP virtual-memory API:
//
//     [ guard: reserved, never mapped ][ mapped pages ............ tensor ][ guard: reserved, never mapped ]
//                                                      ^ right-aligned: the tensor ENDS at the last mapped byte
//
// so a kernel that reads or writes even one element past the end of ANY buffer it was handed (an input, an output, a workspace)
// touches an unmapped page and dies with "Memory access fault by GPU" - in eager mode, on the launch that did it
// (OBMAN_TRACE_LAUNCH=1 makes obman_train_amd/_lib.py name that launch).  The torch caching allocator hides exactly these
// accesses: its blocks sit inside 2 MB / 20 MB segments whose neighbours are mapped.
//
// OBMAN_EFENCE_LEFT=1 left-aligns instead (the tensor STARTS at the first mapped byte: catches accesses BEFORE a buffer).
// The slack between the mapping's other edge and the tensor is filled with a canary byte at allocation and checked when the
// tensor is freed (writes that land in mapped slack: under-runs in the default mode, over-runs in LEFT mode).
// OBMAN_EFENCE_ALIGN (default 16): tensor start alignment in bytes; the right-aligned end is exact for sizes that are a multiple of it.
//
// free() waits for the device, checks the canary and parks the mapping in a quarantine (see efence_free).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
struct Rec {
  char* va;
  size_t va_bytes;
  char* map;
  size_t map_bytes;
  hipMemGenericAllocationHandle_t handle;
  char* user;
  size_t user_bytes;
  size_t seq;
};
std::mutex mu;
std::unordered_map<void*, Rec> live;
std::deque<Rec> quarantine;
size_t bytes_quarantined = 0, quarantine_limit = (size_t)160 << 30, n_released = 0;
size_t gran = 0, user_align = 16;
int left_mode = 0, verbose = 0;
size_t n_alloc = 0, n_free = 0, bytes_live = 0, bytes_peak = 0, canary_hits = 0;
constexpr unsigned char CANARY = 0xA5;
constexpr size_t CANARY_SPAN = 4096;

#define EF_CHECK(expr)                                                                                         \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess) {                                                                                    \
      fprintf(stderr, "[efence] %s failed: %s (%d) at %s:%d\n", #expr, hipGetErrorString(e_), (int)e_, __FILE__, __LINE__); \
      fflush(stderr);                                                                                          \
      abort();                                                                                                 \
    }                                                                                                          \
  } while (0)

int fill_mode = 0, verify_fill = 0;
__global__ void efence_fill_kernel(unsigned char* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = CANARY;
}

void init_once(int device) {
  if (gran) return;
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t g = 0;
  EF_CHECK(hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum));
  gran = g ? g : (size_t)2 << 20;
  if (const char* v = getenv("OBMAN_EFENCE_ALIGN")) user_align = (size_t)atol(v) > 0 ? (size_t)atol(v) : 16;
  if (const char* v = getenv("OBMAN_EFENCE_LEFT")) left_mode = atoi(v);
  if (const char* v = getenv("OBMAN_EFENCE_VERBOSE")) verbose = atoi(v);
  if (const char* v = getenv("OBMAN_EFENCE_QUARANTINE_GB")) quarantine_limit = (size_t)atol(v) << 30;
  if (const char* v = getenv("OBMAN_EFENCE_MEMSET")) fill_mode = atoi(v);    // 1: hipMemsetAsync instead of the fill kernel
  if (const char* v = getenv("OBMAN_EFENCE_VERIFY")) verify_fill = atoi(v);  // 1: read every canary back right after writing it
  fprintf(stderr, "[efence] guard-page allocator active: granularity %zu B, alignment %zu B, %s-aligned tensors\n", gran, user_align,
          left_mode ? "left" : "right");
  fflush(stderr);
}
size_t round_up(size_t n, size_t a) { return (n + a - 1) / a * a; }
}  // namespace

extern "C" {

void* efence_malloc(ssize_t size, int device, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(mu);
  EF_CHECK(hipSetDevice(device));
  init_once(device);
  const size_t user = size > 0 ? (size_t)size : 1;
  const size_t padded = round_up(user, user_align);
  Rec r;
  r.map_bytes = round_up(padded, gran);
  r.va_bytes = r.map_bytes + 2 * gran;
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  void* va = nullptr;
  EF_CHECK(hipMemAddressReserve(&va, r.va_bytes, gran, nullptr, 0));
  r.va = static_cast<char*>(va);
  r.map = r.va + gran;
  EF_CHECK(hipMemCreate(&r.handle, r.map_bytes, &prop, 0));
  EF_CHECK(hipMemMap(r.map, r.map_bytes, 0, r.handle, 0));
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof(acc));
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  EF_CHECK(hipMemSetAccess(r.map, r.map_bytes, &acc, 1));
  r.user = left_mode ? r.map : r.map + (r.map_bytes - padded);
  r.user_bytes = user;
  // canary in the mapped slack next to the tensor (stream-ordered before any use of the tensor on `stream`; other streams only
  // see the tensor after an event the framework records on this one)
  const size_t slack = r.map_bytes - padded;
  const size_t span = slack < CANARY_SPAN ? slack : CANARY_SPAN;
  if (span) {
    char* at = left_mode ? r.user + padded : r.user - span;
    if (fill_mode == 0) {
      efence_fill_kernel<<<(unsigned)((span + 255) / 256), 256, 0, stream>>>(reinterpret_cast<unsigned char*>(at), span);
      EF_CHECK(hipGetLastError());
    } else {
      EF_CHECK(hipMemsetAsync(at, CANARY, span, stream));
    }
    if (verify_fill) {
      EF_CHECK(hipDeviceSynchronize());
      std::vector<unsigned char> host(span);
      EF_CHECK(hipMemcpy(host.data(), at, span, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < span; ++i) bad += host[i] != CANARY;
      if (bad) fprintf(stderr, "[efence] canary fill of %zu bytes at %p did not take: %zu bytes differ (first 0x%02x)\n", span, (void*)at, bad, host[0]);
    }
  }
  r.seq = n_alloc;
  live[r.user] = r;
  ++n_alloc;
  bytes_live += r.map_bytes;
  if (bytes_live > bytes_peak) bytes_peak = bytes_live;
  if (verbose > 1) fprintf(stderr, "[efence] malloc #%zu %zu -> %p (map %p + %zu)\n", r.seq, user, (void*)r.user, (void*)r.map, r.map_bytes);
  return r.user;
}

void efence_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
  (void)size;
  (void)stream;
  if (!ptr) return;
  std::lock_guard<std::mutex> lock(mu);
  auto it = live.find(ptr);
  if (it == live.end()) {
    fprintf(stderr, "[efence] free of unknown pointer %p\n", ptr);
    fflush(stderr);
    abort();
  }
  Rec r = it->second;
  live.erase(it);
  if (verbose > 1) fprintf(stderr, "[efence] free #%zu %p\n", r.seq, ptr);
  if (hipSetDevice(device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    // interpreter shutdown: the runtime is already gone, nothing left to check or unmap
    return;
  }
  const size_t padded = round_up(r.user_bytes, user_align);
  const size_t slack = r.map_bytes - padded;
  const size_t span = slack < CANARY_SPAN ? slack : CANARY_SPAN;
  if (span) {
    std::vector<unsigned char> host(span);
    char* at = left_mode ? r.user + padded : r.user - span;
    EF_CHECK(hipMemcpy(host.data(), at, span, hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0, last = 0;
    for (size_t i = 0; i < span; ++i) {
      if (host[i] != CANARY) {
        if (!bad) first = i;
        last = i;
        ++bad;
      }
    }
    if (bad) {
      ++canary_hits;
      const long rel = left_mode ? (long)(padded + first) : (long)first - (long)span;
      fprintf(stderr, "[efence] CANARY OVERWRITTEN: tensor #%zu %p (%zu bytes), %zu of %zu canary bytes differ, first at offset %ld from the tensor's start, "
              "last %zu bytes later; mapping %p + %zu.  bytes:", r.seq, ptr, r.user_bytes, bad, span, rel, last - first, (void*)r.map, r.map_bytes);
      for (size_t i = first; i <= last && i < first + 48; ++i) fprintf(stderr, " %02x", host[i]);
      fprintf(stderr, "\n");
      for (const auto& kv : live) {  // tensors whose address range is near: an out-of-bounds writer is probably one of them
        const Rec& o = kv.second;
        const long d = (long)(o.map - r.map);
        if (d > -(long)(1 << 20) && d < (long)(1 << 20))
          fprintf(stderr, "[efence]   live neighbour #%zu %p (%zu bytes), mapping %p + %zu (%+ld bytes away)\n", o.seq, (void*)o.user, o.user_bytes,
                  (void*)o.map, o.map_bytes, d);
      }
      fflush(stderr);
      if (!getenv("OBMAN_EFENCE_KEEP_GOING")) abort();
    }
  }
  ++n_free;
  // Freed tensors stay MAPPED in a quarantine (first in, first out) and their addresses are never handed out again while they
  // are in it.  Measured on this stack (ROCm 7.2, gfx950; gpurun_out/r04_efence_dbg64.log): unmapping + releasing right away
  // and mapping the recycled physical page at another address let a later kernel's writes through a STALE translation of the
  // old address land in the new owner's page (the "overwritten canary" held the old neighbour's float data) - a false alarm of
  // the tool, not of the code under test.  With 288 GB of HBM a few steps never reach the limit.
  quarantine.push_back(r);
  bytes_quarantined += r.map_bytes;
  while (bytes_quarantined > quarantine_limit && !quarantine.empty()) {
    const Rec o = quarantine.front();
    quarantine.pop_front();
    EF_CHECK(hipMemUnmap(o.map, o.map_bytes));
    EF_CHECK(hipMemRelease(o.handle));
    EF_CHECK(hipMemAddressFree(o.va, o.va_bytes));
    bytes_quarantined -= o.map_bytes;
    bytes_live -= o.map_bytes;
    ++n_released;
  }
}

// counters for the report: allocations, frees, live, peak mapped bytes, canary hits, mappings released from the quarantine
void efence_stats(size_t* out5) {
  std::lock_guard<std::mutex> lock(mu);
  out5[0] = n_alloc;
  out5[1] = n_free;
  out5[2] = live.size();
  out5[3] = bytes_peak;
  out5[4] = canary_hits;
  out5[5] = n_released;
}
}

// ---- self-test: a kernel that deliberately reads / writes `beyond` bytes past the end (or before the start) of a buffer.
// tools/efence/efence.py --selftest runs it in a child process and expects the GPU memory fault.
__global__ void efence_probe_kernel(const unsigned char* p, long n, long beyond, int write, unsigned* sink) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const long at = beyond >= 0 ? n - 1 + beyond : beyond;
    if (write) const_cast<unsigned char*>(p)[at] = 0x5a;
    else atomicAdd(sink, (unsigned)p[at]);
  }
}
extern "C" int efence_probe(const void* p, long n, long beyond, int write, void* sink, hipStream_t stream) {
  efence_probe_kernel<<<1, 64, 0, stream>>>(static_cast<const unsigned char*>(p), n, beyond, write, static_cast<unsigned*>(sink));
  return (int)hipGetLastError();
}
