// Per-thread staging arena of the host-buffer entry points (round 6).
//
// A host-buffer call is the 1:1 replacement of one reference call on one Frame (plh_orb_search_by_projection_mp, plh_vocab_transform,
// ...).  Rounds 1-5 staged each of them with one hipMalloc + one blocking hipMemcpy per ARRAY on the null stream and a device-wide
// synchronisation -- ten allocations, nine copies and ten hipFree (each a device synchronisation of its own) for a search that runs
// 50 microseconds on the GPU.  Now every calling thread owns ONE device block with a pinned host mirror and a stream of its own:
// the arrays of a call are packed into the mirror, go up in one hipMemcpyAsync, the kernels run on the thread's stream, the
// results come back in one copy and the thread waits for its own stream only.  Nothing is allocated after the first calls of a
// thread have grown the block to their size; two threads (Tracking and LocalMapping call the matchers concurrently) never share
// a buffer or a stream.
#pragma once
#include <algorithm>
#include <vector>

#include "plh_common.h"

namespace plh {

struct StageArena {
  int device = -1;
  hipStream_t stream = nullptr;
  uint8_t* dev = nullptr;    // device block
  uint8_t* host = nullptr;   // pinned mirror of the same size
  size_t cap = 0;
  void release() {
    if (device < 0) return;
    (void)hipSetDevice(device);
    if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
    if (dev) (void)hipFree(dev);
    if (host) (void)hipHostFree(host);
    stream = nullptr; dev = nullptr; host = nullptr; cap = 0; device = -1;
  }
  ~StageArena() { release(); }
};
inline StageArena& stage_arena() {
  static thread_local StageArena a;
  return a;
}

// One host-buffer call's view of the arena: begin(device, upper bound of the bytes) -> in / out / inout / scratch hand out device
// pointers inside the block (256-byte aligned; `in` copies into the pinned mirror right away) -> upload() -> the caller's
// launches on stream() -> download() brings the output range back, waits for the stream and copies to the caller's arrays.
class Stager {
 public:
  static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
  plh_status begin(int device, size_t bytes) {
    StageArena& a = stage_arena();
    ar_ = &a;
    if (a.device != device) a.release();
    PLH_HIP(hipSetDevice(device));
    if (!a.stream) {
      a.device = device;
      PLH_HIP(hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking));
    }
    if (bytes > a.cap) {
      PLH_HIP(hipStreamSynchronize(a.stream));
      if (a.dev) (void)hipFree(a.dev);
      if (a.host) (void)hipHostFree(a.host);
      a.dev = nullptr; a.host = nullptr; a.cap = 0;
      const size_t want = std::max<size_t>(padded(bytes + bytes / 2), (size_t)1 << 20);
      PLH_HIP(hipMalloc((void**)&a.dev, want));
      PLH_HIP(hipHostMalloc((void**)&a.host, want, 0));
      a.cap = want;
    }
    off_ = 0; inHi_ = 0; outLo_ = (size_t)-1; outHi_ = 0; outs_.clear(); ok_ = true;
    return PLH_OK;
  }
  hipStream_t stream() const { return ar_->stream; }
  template <typename T> T* in(const T* src, size_t count) {
    const size_t o = take(count * sizeof(T));
    if (!ok_) return nullptr;
    if (count) memcpy(ar_->host + o, src, count * sizeof(T));
    inHi_ = std::max(inHi_, o + count * sizeof(T));
    return reinterpret_cast<T*>(ar_->dev + o);
  }
  template <typename T> T* in_zero(size_t count) {   // an input that starts as zeros
    const size_t o = take(count * sizeof(T));
    if (!ok_) return nullptr;
    memset(ar_->host + o, 0, count * sizeof(T));
    inHi_ = std::max(inHi_, o + count * sizeof(T));
    return reinterpret_cast<T*>(ar_->dev + o);
  }
  template <typename T> T* out(T* dst, size_t count) {
    const size_t o = take(count * sizeof(T));
    if (!ok_) return nullptr;
    note_out(dst, o, count * sizeof(T));
    return reinterpret_cast<T*>(ar_->dev + o);
  }
  template <typename T> T* inout(T* hostptr, size_t count) {
    const size_t o = take(count * sizeof(T));
    if (!ok_) return nullptr;
    if (count) memcpy(ar_->host + o, hostptr, count * sizeof(T));
    inHi_ = std::max(inHi_, o + count * sizeof(T));
    note_out(hostptr, o, count * sizeof(T));
    return reinterpret_cast<T*>(ar_->dev + o);
  }
  template <typename T> T* scratch(size_t count) {
    const size_t o = take(count * sizeof(T));
    return ok_ ? reinterpret_cast<T*>(ar_->dev + o) : nullptr;
  }
  // the pinned mirror of a device pointer handed out above (the caller fills an input in place instead of copying it twice)
  template <typename T> T* mirror(T* devptr) { return reinterpret_cast<T*>(ar_->host + ((uint8_t*)devptr - ar_->dev)); }
  bool ok() const { return ok_; }
  plh_status upload() {
    if (!ok_) { set_error("staging arena: the call's arrays exceed the bound it was opened with"); return PLH_ERR_INVALID; }
    if (inHi_) PLH_HIP(hipMemcpyAsync(ar_->dev, ar_->host, inHi_, hipMemcpyHostToDevice, ar_->stream));
    return PLH_OK;
  }
  plh_status download() {
    if (outHi_ > outLo_) PLH_HIP(hipMemcpyAsync(ar_->host + outLo_, ar_->dev + outLo_, outHi_ - outLo_, hipMemcpyDeviceToHost, ar_->stream));
    PLH_HIP(hipStreamSynchronize(ar_->stream));
    for (const Out& o : outs_)
      if (o.bytes) memcpy(o.dst, ar_->host + o.off, o.bytes);
    return PLH_OK;
  }

 private:
  struct Out { void* dst; size_t off, bytes; };
  size_t take(size_t bytes) {
    const size_t o = off_;
    off_ += padded(std::max<size_t>(bytes, 1));
    if (off_ > ar_->cap) ok_ = false;
    return o;
  }
  void note_out(void* dst, size_t o, size_t bytes) {
    outs_.push_back(Out{dst, o, bytes});
    outLo_ = std::min(outLo_, o);
    outHi_ = std::max(outHi_, o + bytes);
  }
  StageArena* ar_ = nullptr;
  size_t off_ = 0, inHi_ = 0, outLo_ = (size_t)-1, outHi_ = 0;
  std::vector<Out> outs_;
  bool ok_ = true;
};

}  // namespace plh
