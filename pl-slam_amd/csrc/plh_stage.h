// Per-thread staging arena of the host-buffer entry points (round 6).
//
// A host-buffer call is the 1:1 replacement of one reference call on one Frame (plh_orb_search_by_projection_mp, plh_vocab_transform,
// ...).  Rounds 1-5 staged each of them with one hipMalloc + one blocking hipMemcpy per ARRAY on the null stream and a device-wide
// synchronisation -- about fourteen allocations, eleven copies and fourteen hipFree (each a device synchronisation of its own) for a
// search that runs 50 microseconds on the GPU.  Now every calling thread owns a device block with a pinned host mirror and a stream
// of its own: the arrays of a call are packed into the mirror, go up in one hipMemcpyAsync, the kernels run on the thread's stream,
// the results come back in one copy and the thread waits for its own stream only.  Nothing is allocated once the first calls of a
// thread have grown the block to their size (a call that outgrows it chains a second block and the arena is made one block again
// afterwards); two threads (Tracking and LocalMapping call the matchers concurrently) never share a buffer or a stream.
#pragma once
#include <algorithm>
#include <vector>

#include "plh_common.h"

namespace plh {

struct StageBlock {
  uint8_t* dev = nullptr;    // device block
  uint8_t* host = nullptr;   // pinned mirror of the same size
  size_t cap = 0, used = 0;
  size_t inHi = 0;                        // [0, inHi) of the mirror goes up
  size_t outLo = (size_t)-1, outHi = 0;   // [outLo, outHi) comes back
};
struct StageArena {
  int device = -1;
  hipStream_t stream = nullptr;
  std::vector<StageBlock> blocks;
  void free_blocks() {
    for (StageBlock& b : blocks) {
      if (b.dev) (void)hipFree(b.dev);
      if (b.host) (void)hipHostFree(b.host);
    }
    blocks.clear();
  }
  void release() {
    if (device < 0) return;
    (void)hipSetDevice(device);
    if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
    free_blocks();
    stream = nullptr; device = -1;
  }
  ~StageArena() { release(); }
};
inline StageArena& stage_arena() {
  static thread_local StageArena a;
  return a;
}

// One host-buffer call's view of the arena: begin(device) -> in / in_zero / out / inout / scratch / scratch_zero hand out device
// pointers (256-byte aligned; `in` copies into the pinned mirror right away) -> upload() -> the caller's launches on stream() ->
// download() brings the outputs back, waits for the stream and copies them to the caller's arrays.
class Stager {
 public:
  static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
  // `hint`: bytes the call is going to ask for, when it knows (saves the chained block of a thread's first call)
  plh_status begin(int device, size_t hint = 0) {
    StageArena& a = stage_arena();
    ar_ = &a;
    if (a.device != device) a.release();
    PLH_HIP(hipSetDevice(device));
    if (!a.stream) {
      a.device = device;
      PLH_HIP(hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking));
    }
    size_t sum = 0;
    for (const StageBlock& b : a.blocks) sum += b.cap;
    if (a.blocks.size() > 1 || (!a.blocks.empty() && sum < hint)) {   // the previous call chained blocks (or this one says it is larger): one block again
      PLH_HIP(hipStreamSynchronize(a.stream));
      a.free_blocks();
      if (add_block(std::max(sum, hint)) != PLH_OK) return PLH_ERR_ALLOC;
    }
    for (StageBlock& b : a.blocks) { b.used = 0; b.inHi = 0; b.outLo = (size_t)-1; b.outHi = 0; }
    outs_.clear(); zeros_.clear(); failed_ = false; cur_ = 0;
    return PLH_OK;
  }
  hipStream_t stream() const { return ar_->stream; }
  template <typename T> T* in(const T* src, size_t count, size_t room = 0) {   // `room` >= count elements; the tail starts as zeros
    const size_t n = std::max(count, room) * sizeof(T);
    Slot s = take(n);
    if (!s.dev) return nullptr;
    if (count && src) memcpy(s.host, src, count * sizeof(T));
    if (n > count * sizeof(T) || !src) memset(s.host + (src ? count * sizeof(T) : 0), 0, n - (src ? count * sizeof(T) : 0));
    s.b->inHi = std::max(s.b->inHi, s.off + n);
    return reinterpret_cast<T*>(s.dev);
  }
  template <typename T> T* out(T* dst, size_t count, size_t room = 0) {   // count elements come back to dst
    Slot s = take(std::max(count, room) * sizeof(T));
    if (!s.dev) return nullptr;
    note_out(s, dst, count * sizeof(T));
    return reinterpret_cast<T*>(s.dev);
  }
  template <typename T> T* inout(T* hostptr, size_t count, size_t room = 0) {
    const size_t n = std::max(count, room) * sizeof(T);
    Slot s = take(n);
    if (!s.dev) return nullptr;
    if (count) memcpy(s.host, hostptr, count * sizeof(T));
    if (n > count * sizeof(T)) memset(s.host + count * sizeof(T), 0, n - count * sizeof(T));
    s.b->inHi = std::max(s.b->inHi, s.off + n);
    note_out(s, hostptr, count * sizeof(T));
    return reinterpret_cast<T*>(s.dev);
  }
  template <typename T> T* scratch(size_t count) { return reinterpret_cast<T*>(take(count * sizeof(T)).dev); }
  template <typename T> T* scratch_zero(size_t count) {   // device-only, cleared on the stream by upload()
    Slot s = take(count * sizeof(T));
    if (s.dev) zeros_.push_back(Zero{s.dev, count * sizeof(T)});
    return reinterpret_cast<T*>(s.dev);
  }
  // bring `count` elements at device pointer `d` (handed out above) back to `dst` at download()
  template <typename T> void fetch(T* dst, const T* d, size_t count) {
    for (StageBlock& b : ar_->blocks)
      if ((const uint8_t*)d >= b.dev && (const uint8_t*)d < b.dev + b.cap) {
        Slot s{&b, (size_t)((const uint8_t*)d - b.dev), const_cast<uint8_t*>((const uint8_t*)d), b.host + ((const uint8_t*)d - b.dev)};
        note_out(s, dst, count * sizeof(T));
        return;
      }
    failed_ = true;
  }
  plh_status upload() {
    if (failed_) { set_error("staging arena: allocation failed"); return PLH_ERR_ALLOC; }
    for (StageBlock& b : ar_->blocks)
      if (b.inHi) PLH_HIP(hipMemcpyAsync(b.dev, b.host, b.inHi, hipMemcpyHostToDevice, ar_->stream));
    for (const Zero& z : zeros_) PLH_HIP(hipMemsetAsync(z.p, 0, z.bytes, ar_->stream));
    return PLH_OK;
  }
  plh_status download() {
    if (failed_) { set_error("staging arena: allocation failed"); return PLH_ERR_ALLOC; }
    for (StageBlock& b : ar_->blocks)
      if (b.outHi > b.outLo) PLH_HIP(hipMemcpyAsync(b.host + b.outLo, b.dev + b.outLo, b.outHi - b.outLo, hipMemcpyDeviceToHost, ar_->stream));
    PLH_HIP(hipStreamSynchronize(ar_->stream));
    for (const Out& o : outs_)
      if (o.bytes) memcpy(o.dst, o.src, o.bytes);
    return PLH_OK;
  }

 private:
  struct Slot { StageBlock* b; size_t off; uint8_t* dev; uint8_t* host; };
  struct Out { void* dst; const uint8_t* src; size_t bytes; };
  struct Zero { void* p; size_t bytes; };
  plh_status add_block(size_t bytes) {
    StageBlock b;
    b.cap = std::max<size_t>(padded(bytes + bytes / 2), (size_t)1 << 20);
    if (hipMalloc((void**)&b.dev, b.cap) != hipSuccess || hipHostMalloc((void**)&b.host, b.cap, 0) != hipSuccess) {
      (void)hipGetLastError();
      if (b.dev) (void)hipFree(b.dev);
      set_error("staging arena: cannot allocate %zu bytes", b.cap);
      return PLH_ERR_ALLOC;
    }
    ar_->blocks.push_back(b);
    return PLH_OK;
  }
  Slot take(size_t bytes) {
    const size_t n = padded(std::max<size_t>(bytes, 1));
    if (failed_) return Slot{nullptr, 0, nullptr, nullptr};
    while (cur_ < ar_->blocks.size() && ar_->blocks[cur_].used + n > ar_->blocks[cur_].cap) cur_++;
    if (cur_ >= ar_->blocks.size()) {
      size_t sum = 0;
      for (const StageBlock& b : ar_->blocks) sum += b.cap;
      if (add_block(std::max(n, sum)) != PLH_OK) { failed_ = true; return Slot{nullptr, 0, nullptr, nullptr}; }
      cur_ = ar_->blocks.size() - 1;
    }
    StageBlock& b = ar_->blocks[cur_];
    Slot s{&b, b.used, b.dev + b.used, b.host + b.used};
    b.used += n;
    return s;
  }
  void note_out(const Slot& s, void* dst, size_t bytes) {
    outs_.push_back(Out{dst, s.host, bytes});
    s.b->outLo = std::min(s.b->outLo, s.off);
    s.b->outHi = std::max(s.b->outHi, s.off + bytes);
  }
  StageArena* ar_ = nullptr;
  size_t cur_ = 0;
  std::vector<Out> outs_;
  std::vector<Zero> zeros_;
  bool failed_ = false;
};

}  // namespace plh
