// ga_comm.cpp -- the one exchange step of the sharded render: the sum of the destination bus across GPUs.
// RCCL is loaded at run time (dlopen) by the first ga_comm_* call, so libgraphaudio_hip.so itself depends on the HIP runtime
// only and single-GPU hosts never load a collective library.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include <chrono>
#include <thread>

#include "ga_engine.hpp"

// RCCL is loaded with dlopen, so its declarations are restated below.  Where the header is present at build time the restated
// constants and the by-value id are checked against it (a silent mismatch would corrupt the collective, not fail it).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert(NCCL_UNIQUE_ID_BYTES == GA_COMM_ID_BYTES, "include/graphaudio_hip.h: GA_COMM_ID_BYTES != NCCL_UNIQUE_ID_BYTES");
static_assert(sizeof(ncclUniqueId) == GA_COMM_ID_BYTES, "ncclUniqueId is passed by value as GA_COMM_ID_BYTES bytes");
static_assert((int)ncclSuccess == 0 && (int)ncclInProgress == 7, "restated ncclResult_t values");
static_assert((int)ncclFloat32 == 7 && (int)ncclSum == 0, "restated ncclDataType_t / ncclRedOp_t values");
#endif

namespace ga {

namespace {
constexpr int kNcclSuccess = 0, kNcclInProgress = 7, kNcclFloat32 = 7, kNcclSum = 0;
struct UniqueId {
  char internal[GA_COMM_ID_BYTES];
};
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
  int (*CommAbort)(void*) = nullptr;                 // (optional: older libraries)
  int (*CommGetAsyncError)(void*, int*) = nullptr;   // (optional)
  int (*CommCount)(void*, int*) = nullptr;           // (optional)
  int (*CommUserRank)(void*, int*) = nullptr;        // (optional)
  const char* (*GetErrorString)(int) = nullptr;
  std::string error;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      r.error = std::string("cannot load RCCL (librccl.so.1): ") + dlerror();
      return;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.Reduce = (decltype(r.Reduce))dlsym(r.lib, "ncclReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");
    r.CommGetAsyncError = (decltype(r.CommGetAsyncError))dlsym(r.lib, "ncclCommGetAsyncError");
    r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.lib, "ncclCommUserRank");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Reduce || !r.GetErrorString) r.error = "RCCL library lacks an expected symbol";
  });
  return r;
}
void need(Rccl& r) {
  if (!r.error.empty()) fail(GA_ERR_UNSUPPORTED, r.error);
}
void check(Rccl& r, int rc, const char* what) {
  if (rc != 0) fail(GA_ERR_DEVICE, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}
}  // namespace

void commUniqueId(void* out) {
  if (!out) fail(GA_ERR_INVALID_ARGUMENT, "null pointer");
  Rccl& r = rccl();
  need(r);
  UniqueId id;
  check(r, r.GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out, &id, sizeof(id));
}

void Context::commInit(const void* id, int nRanks, int rank) {
  if (nRanks < 1 || rank < 0 || rank >= nRanks) fail(GA_ERR_OUT_OF_RANGE, "rank / n_ranks");
  if (comm || commRanks) fail(GA_ERR_INVALID_OPERATION, "the context already belongs to a communicator");
  GA_HIP(hipSetDevice(device));
  if (nRanks > 1) {
    if (!id) fail(GA_ERR_INVALID_ARGUMENT, "null communicator id");
    Rccl& r = rccl();
    need(r);
    UniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    void* c = nullptr;
    check(r, r.CommInitRank(&c, nRanks, uid, rank), "ncclCommInitRank");
    comm = c;
  }
  commRanks = nRanks;
  commRank = rank;
}

// ncclCommCount / ncclCommUserRank of the live communicator (include/graphaudio_hip.h, ga_comm_info)
void Context::commInfo(int* nRanks, int* rank, int* usesRccl) {
  if (!nRanks || !rank || !usesRccl) fail(GA_ERR_INVALID_ARGUMENT, "null pointer");
  *nRanks = 1;
  *rank = 0;
  *usesRccl = 0;
  if (!comm) return;
  Rccl& r = rccl();
  if (!r.CommCount || !r.CommUserRank) fail(GA_ERR_UNSUPPORTED, "the RCCL library lacks ncclCommCount / ncclCommUserRank");
  check(r, r.CommCount(comm, nRanks), "ncclCommCount");
  check(r, r.CommUserRank(comm, rank), "ncclCommUserRank");
  *usesRccl = 1;
}

// A rank that cannot take part in a collective its peers have already enqueued (its render failed) must not leave them waiting:
// the communicator is aborted -- the peers' collectives then end with an error, which their commWait() turns into an error code.
void Context::commAbort() {
  if (!comm) return;
  Rccl& r = rccl();
  if (r.CommAbort) (void)r.CommAbort(comm);
  else if (r.CommDestroy) (void)r.CommDestroy(comm);
  comm = nullptr;
  commDead = true;
}

// Wait for the stream of a context that has collectives in flight: a peer that died or aborted shows up as an asynchronous
// error of the communicator (or, at worst, as the time limit of option "comm_timeout_s"), not as a wait that never ends.
void Context::commWait() {
  if (!comm) {
    GA_HIP(hipStreamSynchronize(stream));
    return;
  }
  Rccl& r = rccl();
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(stream);
    if (q == hipSuccess) return;
    if (q != hipErrorNotReady) GA_HIP(q);
    (void)hipGetLastError();
    int async = kNcclSuccess;
    if (r.CommGetAsyncError && r.CommGetAsyncError(comm, &async) == kNcclSuccess && async != kNcclSuccess && async != kNcclInProgress) {
      const std::string what = r.GetErrorString ? r.GetErrorString(async) : "RCCL error";
      commAbort();
      fail(GA_ERR_DEVICE, "the communicator failed (a peer rank aborted or died): " + what);
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > commTimeoutS) {
      commAbort();
      fail(GA_ERR_DEVICE, "timed out waiting for the sharded render's collective (option comm_timeout_s)");
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

void Context::commDestroy() {
  if (comm) {
    try {
      if (stream) commWait();
    } catch (...) {   // (commWait aborted the communicator)
    }
  }
  if (comm) {
    Rccl& r = rccl();
    if (r.CommDestroy) (void)r.CommDestroy(comm);
    comm = nullptr;
  }
  commDead = false;
  commRanks = 0;
  commRank = 0;
  if (reduceBuf) {
    dfree(reduceBuf, reduceBytes);
    reduceBuf = nullptr;
    reduceBytes = 0;
  }
}

// render this rank's share into device memory, sum over the ranks, hand the result to the root's caller -- all on the
// context's stream, so that with option "async" consecutive calls pipeline (the host plans call k + 1 while the device
// renders and reduces call k)
void Context::renderReduce(float* const* out, int channels, int64_t frames, int64_t start, int root) {
  if (commRanks < 1) fail(GA_ERR_INVALID_OPERATION, "ga_comm_init has not been called on this context");
  if (commDead) fail(GA_ERR_INVALID_OPERATION, "the communicator was aborted after an error; ga_comm_destroy and ga_comm_init again");
  if (root < 0 || root >= commRanks) fail(GA_ERR_OUT_OF_RANGE, "root");
  if (channels < 1 || channels > 32) fail(GA_ERR_OUT_OF_RANGE, "channelIndex");
  if (frames <= 0) fail(GA_ERR_OUT_OF_RANGE, "Frame count must be positive.");
  if (start < 0) fail(GA_ERR_OUT_OF_RANGE, "Start index must be non-negative.");
  const bool isRoot = commRank == root;
  if (isRoot) {
    if (!out) fail(GA_ERR_INVALID_ARGUMENT, "Output buffer must have at least one channel.");
    for (int ch = 0; ch < channels; ch++)
      if (!out[ch]) fail(GA_ERR_INVALID_ARGUMENT, "Channel buffer is null.");
  }
  GA_HIP(hipSetDevice(device));
  const size_t need_ = (size_t)channels * (size_t)frames * sizeof(float);
  if (reduceBytes < need_) {
    flushHandOver();   // (a pending hand-over reads the buffer that is about to go)
    if (reduceBuf) {
      GA_HIP(hipStreamSynchronize(stream));
      dfree(reduceBuf, reduceBytes);
    }
    reduceBytes = need_ + need_ / 8;
    reduceBuf = (float*)dalloc(reduceBytes);
  }
  float* rows[32];
  for (int ch = 0; ch < channels; ch++) rows[ch] = reduceBuf + (size_t)ch * frames;
  const bool callerAsync = asyncMode;
  asyncMode = true;   // everything below is ordered on the stream; wait once at the end unless the caller pipelines
  try {
    render(rows, channels, frames, 0, true);
    if (comm) {
      Rccl& r = rccl();
      check(r, r.Reduce(reduceBuf, reduceBuf, (size_t)channels * frames, kNcclFloat32, kNcclSum, root, comm, stream), "ncclReduce");
    }
    if (isRoot) {
      // page-locked rows of an asynchronous caller: the sum crosses PCIe inside the next step's pre-mix launch (Context::pendingHandOver)
      float* devAlias[32];
      bool locked = callerAsync && hostDefer && ownStream && (frames % 4) == 0;   // (never on a caller-supplied stream, see Context::render)
      for (int ch = 0; ch < channels && locked; ch++) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, out[ch] + start) != hipSuccess || at.type != hipMemoryTypeHost || !at.devicePointer ||
            (((uintptr_t)at.devicePointer) & 15)) {
          (void)hipGetLastError();
          locked = false;
        } else {
          devAlias[ch] = (float*)at.devicePointer;
        }
      }
      if (locked) {
        for (int ch = 0; ch < channels; ch++) pendingHandOver.push_back(HandOver{rows[ch], devAlias[ch], out[ch] + start, frames});
      } else if (callerAsync && ownStream && hostCopyStream) {
        handOverToHost(rows, out, channels, start, frames);   // (leaves on the copy stream: ga_engine.hpp)
      } else {
        for (int ch = 0; ch < channels; ch++)
          GA_HIP(hipMemcpyAsync(out[ch] + start, rows[ch], (size_t)frames * sizeof(float), hipMemcpyDeviceToHost, stream));
      }
    }
  } catch (...) {
    asyncMode = callerAsync;
    commAbort();   // the peers have (or will have) this step's collective on their streams: do not leave them waiting
    throw;
  }
  asyncMode = callerAsync;
  if (!callerAsync) {
    flushHandOver();
    commWait();
    harvestProfile(true);
  }
}

}  // namespace ga
