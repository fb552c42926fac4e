// wan_sp_*: a library-owned RCCL communicator for temporal sequence parallelism (SURVEY.md section 8b `wan_sp_init(rank, nranks,
// ncclUniqueId)`).  One communicator per process / GPU, one side HIP stream, one completion event per gather slot:
//
//   wan_sp_gather_begin  record an event on the compute stream -> the side stream waits for it -> ncclAllGather on the side
//                        stream -> record the slot's completion event.  Returns at once: the compute stream keeps going.
//   wan_sp_gather_wait   the compute stream waits for the slot's completion event.
//
// Both have the wan_gather_begin_fn / wan_gather_wait_fn signatures: put them (with the wan_sp* as `user`) into wan_sp_info and
// wan_dit_forward drives the K / V^T all-gathers over xGMI without leaving the library.  The alternative that ships as the
// default is the host callback pair of wan2gp_amd/sp.py (torch.distributed owns the communicator).
//
// RCCL is bound at run time (dlopen): a process that already carries an RCCL (PyTorch bundles its own librccl.so) keeps ONE
// instance -- two copies of the library in a process each run their own proxy threads and topology discovery.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {

// the slice of <rccl/rccl.h> this file uses (types only; the functions are looked up with dlsym)
constexpr int UNIQUE_ID_BYTES = 128;
struct UniqueId { char internal[UNIQUE_ID_BYTES]; };
typedef void* Comm;
typedef int Result;        // ncclResult_t: 0 = ncclSuccess
constexpr int kNcclInt8 = 0;  // ncclDataType_t ncclInt8 / ncclChar

struct Api {
  Result (*GetUniqueId)(UniqueId*) = nullptr;
  Result (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  Result (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  Result (*CommDestroy)(Comm) = nullptr;
  const char* (*GetErrorString)(Result) = nullptr;
  // the all-to-all of WAN_SP_ULYSSES: grouped point-to-point (ncclAllToAll is an RCCL extension this file does not rely on)
  Result (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  Result (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  Result (*GroupStart)() = nullptr;
  Result (*GroupEnd)() = nullptr;
  bool ok = false, p2p = false;
};

const Api& api() {
  static const Api a = [] {
    Api r;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);             // the instance the process already has (PyTorch's)
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);           // none loaded: the system's
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return r;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy && r.GetErrorString;
    r.p2p = r.ok && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
    return r;
  }();
  return a;
}

#define WAN_CHECK_NCCL(expr)                                                                        \
  do {                                                                                              \
    const Result _r = (expr);                                                                       \
    if (_r != 0) {                                                                                  \
      wan_set_error("%s failed: %s (%s:%d)", #expr, api().GetErrorString(_r), __FILE__, __LINE__); \
      return 2;                                                                                     \
    }                                                                                               \
  } while (0)

constexpr int kSlots = 4 * WAN_SP_MAX_CHUNKS + 1;   // the k, v^T, q and o chunks of a block (C each: k c, v^T C + c, q 2 C + c, o 3 C + c); the last: wan_sp_all_gather

}  // namespace

struct wan_sp {
  Comm comm = nullptr;
  int rank = 0, world = 1;
  hipStream_t side = nullptr;
  hipEvent_t ready = nullptr;          // "everything the collective reads has been enqueued" on the compute stream
  hipEvent_t done[kSlots] = {};        // per gather slot: the collective has finished
  bool pending[kSlots] = {};
};

extern "C" void wan_sp_destroy(wan_sp* s);

// 128 bytes identifying a new communicator: rank 0 creates it, the host runtime hands it to every rank (any broadcast will do)
extern "C" int wan_sp_unique_id(void* id128) {
  WAN_REQUIRE(id128 != nullptr, "wan_sp_unique_id: null pointer");
  WAN_REQUIRE(api().ok, "wan_sp: no RCCL library could be loaded (librccl.so)");
  UniqueId id;
  WAN_CHECK_NCCL(api().GetUniqueId(&id));
  memcpy(id128, id.internal, UNIQUE_ID_BYTES);
  return 0;
}

// Collective: every rank of the group calls it with the same id.  Binds to the calling thread's current HIP device.
extern "C" int wan_sp_init(wan_sp** out, int rank, int nranks, const void* id128) {
  WAN_REQUIRE(out && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "wan_sp_init: bad arguments (rank %d of %d)", rank, nranks);
  WAN_REQUIRE(api().ok, "wan_sp: no RCCL library could be loaded (librccl.so)");
  wan_sp* s = new wan_sp();
  s->rank = rank;
  s->world = nranks;
  UniqueId id;
  memcpy(id.internal, id128, UNIQUE_ID_BYTES);
  const Result r = api().CommInitRank(&s->comm, nranks, id, rank);
  if (r != 0) {
    wan_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, api().GetErrorString(r));
    delete s;
    return 2;
  }
  hipError_t e = hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ready, hipEventDisableTiming);
  for (int i = 0; i < kSlots && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&s->done[i], hipEventDisableTiming);
  if (e != hipSuccess) {
    wan_set_error("wan_sp_init: stream / event creation failed: %s", hipGetErrorString(e));
    wan_sp_destroy(s);  // the communicator and whatever was created: a failed init leaves nothing behind
    return 2;
  }
  *out = s;
  return 0;
}

extern "C" void wan_sp_destroy(wan_sp* s) {
  if (!s) return;
  if (s->side) (void)hipStreamSynchronize(s->side);
  if (s->comm && api().ok) (void)api().CommDestroy(s->comm);
  for (hipEvent_t ev : s->done)
    if (ev) (void)hipEventDestroy(ev);
  if (s->ready) (void)hipEventDestroy(s->ready);
  if (s->side) (void)hipStreamDestroy(s->side);
  delete s;
}

// wan_gather_begin_fn: recv[world][bytes] <- all ranks' send[bytes]; `which` selects the completion slot (0 = K, 1 = V^T, ...)
extern "C" int wan_sp_gather_begin(void* user, int which, const void* send, void* recv, int64_t bytes, void* stream) {
  wan_sp* s = static_cast<wan_sp*>(user);
  WAN_REQUIRE(s && send && recv && bytes > 0 && which >= 0 && which < kSlots, "wan_sp_gather_begin: bad arguments (slot %d)", which);
  WAN_REQUIRE(!s->pending[which], "wan_sp_gather_begin: slot %d still has a gather in flight (missing wan_sp_gather_wait)", which);
  WAN_CHECK_HIP(hipEventRecord(s->ready, as_stream(stream)));
  WAN_CHECK_HIP(hipStreamWaitEvent(s->side, s->ready, 0));
  WAN_CHECK_NCCL(api().AllGather(send, recv, (size_t)bytes, kNcclInt8, s->comm, s->side));
  WAN_CHECK_HIP(hipEventRecord(s->done[which], s->side));
  s->pending[which] = true;
  return 0;
}

// wan_gather_begin_fn signature, all-to-all semantics (WAN_SP_ULYSSES): chunk j of send[world][bytes] goes to rank j, chunk i of
// recv[world][bytes] comes from rank i.  One group of ncclSend / ncclRecv pairs on the side stream (every pair uses its own xGMI
// link: point-to-point is what the fabric is); the rank's own chunk is a device-to-device copy.
extern "C" int wan_sp_a2a_begin(void* user, int which, const void* send, void* recv, int64_t bytes, void* stream) {
  wan_sp* s = static_cast<wan_sp*>(user);
  WAN_REQUIRE(s && send && recv && bytes > 0 && which >= 0 && which < kSlots, "wan_sp_a2a_begin: bad arguments (slot %d)", which);
  WAN_REQUIRE(api().p2p, "wan_sp_a2a_begin: the loaded RCCL has no ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
  WAN_REQUIRE(!s->pending[which], "wan_sp_a2a_begin: slot %d still has an exchange in flight (missing wan_sp_gather_wait)", which);
  WAN_CHECK_HIP(hipEventRecord(s->ready, as_stream(stream)));
  WAN_CHECK_HIP(hipStreamWaitEvent(s->side, s->ready, 0));
  const char* sb = static_cast<const char*>(send);
  char* rb = static_cast<char*>(recv);
  WAN_CHECK_HIP(hipMemcpyAsync(rb + (size_t)s->rank * bytes, sb + (size_t)s->rank * bytes, (size_t)bytes, hipMemcpyDeviceToDevice, s->side));
  WAN_CHECK_NCCL(api().GroupStart());
  for (int p = 0; p < s->world; ++p) {
    if (p == s->rank) continue;
    const Result a = api().Send(sb + (size_t)p * bytes, (size_t)bytes, kNcclInt8, p, s->comm, s->side);
    const Result b = a == 0 ? api().Recv(rb + (size_t)p * bytes, (size_t)bytes, kNcclInt8, p, s->comm, s->side) : a;
    if (b != 0) {
      (void)api().GroupEnd();
      wan_set_error("wan_sp_a2a_begin: ncclSend / ncclRecv with rank %d failed: %s", p, api().GetErrorString(b));
      return 2;
    }
  }
  WAN_CHECK_NCCL(api().GroupEnd());
  WAN_CHECK_HIP(hipEventRecord(s->done[which], s->side));
  s->pending[which] = true;
  return 0;
}

// wan_gather_wait_fn: `stream` waits for slot `which` (no host wait)
extern "C" int wan_sp_gather_wait(void* user, int which, void* stream) {
  wan_sp* s = static_cast<wan_sp*>(user);
  WAN_REQUIRE(s && which >= 0 && which < kSlots, "wan_sp_gather_wait: bad arguments (slot %d)", which);
  if (!s->pending[which]) return 0;
  WAN_CHECK_HIP(hipStreamWaitEvent(as_stream(stream), s->done[which], 0));
  s->pending[which] = false;
  return 0;
}

// an all-gather ordered on `stream` on both sides (the head's token-major output once per forward)
extern "C" int wan_sp_all_gather(wan_sp* s, const void* send, void* recv, int64_t bytes, void* stream) {
  if (int rc = wan_sp_gather_begin(s, kSlots - 1, send, recv, bytes, stream)) return rc;
  return wan_sp_gather_wait(s, kSlots - 1, stream);
}
