// Multi-GPU collection of the front end's fixed-stride records over RCCL (xGMI inside a node), in the C ABI so that a C++
// host needs nothing but this library (frames are independent, every rank extracts its own batch, the only
// exchange is the gather of the results).
//   plh_comm_unique_id / plh_comm_create / plh_comm_wrap / plh_comm_destroy   communicator = one ncclComm_t per process / GPU
//   plh_gather_records                                                        all-gather or gather-to-root of n record blocks,
//                                                                             one grouped RCCL launch on the caller's stream
// RCCL is bound at run time (dlopen of the copy already in the process -- e.g. PyTorch's -- else librccl.so.1), so the library
// has no link-time dependency on it and single-GPU hosts never load it.  The emulator build (tests/hipemu) has no RCCL: its
// one-rank communicator copies the blocks, which is what a one-rank gather is.
#include "plh_common.h"

#if !defined(HIPEMU)
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#endif

struct plh_comm {
  int rank = 0, world = 1, device = 0, version = 0;
  bool owned = false;
#if !defined(HIPEMU)
  ncclComm_t comm = nullptr;
#endif
};

using namespace plh;

#if !defined(HIPEMU)
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;   // RCCL extension
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl* rccl_bind(Rccl& r) {
  // RTLD_NOLOAD first: reuse the RCCL the process already holds (two copies would each own their own communicators)
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !r.lib; pass++)
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (r.lib) break;
    }
  if (!r.lib) return nullptr;
#define BIND(field, sym) *(void**)(&r.field) = dlsym(r.lib, sym)
  BIND(GetUniqueId, "ncclGetUniqueId"); BIND(CommInitRank, "ncclCommInitRank"); BIND(CommDestroy, "ncclCommDestroy");
  BIND(CommCount, "ncclCommCount"); BIND(CommUserRank, "ncclCommUserRank"); BIND(AllGather, "ncclAllGather");
  BIND(Gather, "ncclGather"); BIND(Send, "ncclSend"); BIND(Recv, "ncclRecv"); BIND(GroupStart, "ncclGroupStart");
  BIND(GroupEnd, "ncclGroupEnd"); BIND(GetVersion, "ncclGetVersion"); BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) {
    r.lib = nullptr;
    return nullptr;
  }
  return &r;
}
// bound once, by whichever thread comes first; the others wait for the finished table (never a half-bound one)
Rccl* rccl() {
  static Rccl r;
  static Rccl* bound = nullptr;
  static std::once_flag once;
  std::call_once(once, []() { bound = rccl_bind(r); });
  return bound;
}
#define PLH_NCCL(R, call)                                                                                   \
  do {                                                                                                      \
    ncclResult_t e__ = (call);                                                                              \
    if (e__ != ncclSuccess) {                                                                               \
      set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, (R)->GetErrorString ? (R)->GetErrorString(e__) : "RCCL error"); \
      return PLH_ERR_HIP;                                                                                   \
    }                                                                                                       \
  } while (0)
}  // namespace
#endif

extern "C" {

plh_status plh_comm_unique_id(uint8_t id[PLH_COMM_ID_BYTES]) {
  if (!id) return PLH_ERR_INVALID;
#if defined(HIPEMU)
  memset(id, 0, PLH_COMM_ID_BYTES);
  return PLH_OK;
#else
  static_assert(PLH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  Rccl* R = rccl();
  if (!R) { set_error("plh_comm_unique_id: RCCL (librccl.so.1) cannot be loaded"); return PLH_ERR_NO_DEVICE; }
  ncclUniqueId u;
  PLH_NCCL(R, R->GetUniqueId(&u));
  memcpy(id, u.internal, PLH_COMM_ID_BYTES);
  return PLH_OK;
#endif
}

plh_status plh_comm_create(const uint8_t id[PLH_COMM_ID_BYTES], int rank, int world, int device, plh_comm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) { set_error("plh_comm_create: invalid argument"); return PLH_ERR_INVALID; }
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  plh_comm* c = new plh_comm();
  c->rank = rank; c->world = world; c->device = device; c->owned = true;
#if defined(HIPEMU)
  if (world != 1) { delete c; set_error("plh_comm_create: the emulator build has no RCCL (world must be 1)"); return PLH_ERR_INVALID; }
#else
  Rccl* R = rccl();
  if (!R) { delete c; set_error("plh_comm_create: RCCL (librccl.so.1) cannot be loaded"); return PLH_ERR_NO_DEVICE; }
  if (hipSetDevice(device) != hipSuccess) { delete c; set_error("plh_comm_create: hipSetDevice(%d) failed", device); return PLH_ERR_NO_DEVICE; }
  ncclUniqueId u;
  memcpy(u.internal, id, PLH_COMM_ID_BYTES);
  ncclResult_t e = R->CommInitRank(&c->comm, world, u, rank);
  if (e != ncclSuccess) {
    set_error("plh_comm_create: ncclCommInitRank -> %s", R->GetErrorString ? R->GetErrorString(e) : "RCCL error");
    delete c;
    return PLH_ERR_HIP;
  }
  if (R->GetVersion) R->GetVersion(&c->version);
#endif
  *out = c;
  return PLH_OK;
}

plh_status plh_comm_wrap(void* nccl_comm, int rank, int world, plh_comm** out) {
  if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return PLH_ERR_INVALID;
#if defined(HIPEMU)
  set_error("plh_comm_wrap: the emulator build has no RCCL");
  return PLH_ERR_INVALID;
#else
  Rccl* R = rccl();
  if (!R) { set_error("plh_comm_wrap: RCCL (librccl.so.1) cannot be loaded"); return PLH_ERR_NO_DEVICE; }
  plh_comm* c = new plh_comm();
  c->comm = (ncclComm_t)nccl_comm; c->rank = rank; c->world = world; c->owned = false;
  if (R->GetVersion) R->GetVersion(&c->version);
  *out = c;
  return PLH_OK;
#endif
}

plh_status plh_comm_destroy(plh_comm* c) {
  if (!c) return PLH_OK;
#if !defined(HIPEMU)
  if (c->owned && c->comm) {
    Rccl* R = rccl();
    if (R) (void)R->CommDestroy(c->comm);
  }
#endif
  delete c;
  return PLH_OK;
}

plh_status plh_comm_info(const plh_comm* c, int* rank, int* world, int* rccl_version) {
  if (!c) return PLH_ERR_INVALID;
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (rccl_version) *rccl_version = c->version;
  return PLH_OK;
}

plh_status plh_gather_records(plh_comm* c, const plh_gather_block* blocks, int nblocks, int root, void* stream) {
  if (!c || !blocks || nblocks <= 0 || root < -1 || root >= c->world) { set_error("plh_gather_records: invalid argument"); return PLH_ERR_INVALID; }
  const bool receives = root < 0 || root == c->rank;
  for (int i = 0; i < nblocks; i++)
    if (!blocks[i].send || (receives && !blocks[i].recv)) { set_error("plh_gather_records: block %d has a null buffer", i); return PLH_ERR_INVALID; }
  hipStream_t s = (hipStream_t)stream;
#if defined(HIPEMU)
  for (int i = 0; i < nblocks; i++)
    if (blocks[i].recv != blocks[i].send) PLH_HIP(hipMemcpyAsync(blocks[i].recv, blocks[i].send, blocks[i].bytes, hipMemcpyDeviceToDevice, s));
  return PLH_OK;
#else
  Rccl* R = rccl();
  if (!R) { set_error("plh_gather_records: RCCL (librccl.so.1) cannot be loaded"); return PLH_ERR_NO_DEVICE; }
  // one grouped launch for all blocks of the sub-batch (counts, keypoints, descriptors, keylines, LBD, ...).  A failure
  // inside the group must not leave it open (every later RCCL call of this thread, PyTorch's included, would queue behind it
  // and never launch): the first error is remembered, the group is closed, then it is reported.
  PLH_NCCL(R, R->GroupStart());
  ncclResult_t first = ncclSuccess;
  const char* what = "";
  auto note = [&](ncclResult_t e, const char* w) { if (e != ncclSuccess && first == ncclSuccess) { first = e; what = w; } };
  for (int i = 0; i < nblocks && first == ncclSuccess; i++) {
    const plh_gather_block& b = blocks[i];
    if (root < 0) {
      note(R->AllGather(b.send, b.recv, b.bytes, ncclUint8, c->comm, s), "ncclAllGather");
    } else if (R->Gather) {
      note(R->Gather(b.send, b.recv, b.bytes, ncclUint8, root, c->comm, s), "ncclGather");
    } else {   // plain NCCL API: the root posts one receive per rank, everybody sends
      if (c->rank == root)
        for (int r = 0; r < c->world && first == ncclSuccess; r++)
          note(R->Recv((char*)b.recv + (size_t)r * b.bytes, b.bytes, ncclUint8, r, c->comm, s), "ncclRecv");
      if (first == ncclSuccess) note(R->Send(b.send, b.bytes, ncclUint8, root, c->comm, s), "ncclSend");
    }
  }
  const ncclResult_t eEnd = R->GroupEnd();
  if (first == ncclSuccess) note(eEnd, "ncclGroupEnd");
  if (first != ncclSuccess) {
    set_error("plh_gather_records: %s -> %s", what, R->GetErrorString ? R->GetErrorString(first) : "RCCL error");
    return PLH_ERR_HIP;
  }
  return PLH_OK;
#endif
}

}  // extern "C"
