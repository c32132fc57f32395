// ga_comm.cpp -- the one exchange step of the sharded render: the sum of the destination bus across GPUs.
// RCCL is loaded at run time (dlopen) by the first ga_comm_* call, so libgraphaudio_hip.so itself depends on the HIP runtime
// only and single-GPU hosts never load a collective library.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "ga_engine.hpp"

namespace ga {

namespace {
// the few RCCL entry points used (rccl/rccl.h: NCCL_UNIQUE_ID_BYTES = 128, ncclFloat32 = 7, ncclSum = 0, ncclSuccess = 0)
struct UniqueId {
  char internal[GA_COMM_ID_BYTES];
};
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
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

void Context::commDestroy() {
  if (comm) {
    if (stream) (void)hipStreamSynchronize(stream);
    Rccl& r = rccl();
    if (r.CommDestroy) (void)r.CommDestroy(comm);
    comm = nullptr;
  }
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
  flushHandOver();
  const size_t need_ = (size_t)channels * (size_t)frames * sizeof(float);
  if (reduceBytes < need_) {
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
      check(r, r.Reduce(reduceBuf, reduceBuf, (size_t)channels * frames, /*ncclFloat32*/ 7, /*ncclSum*/ 0, root, comm, stream), "ncclReduce");
    }
    if (isRoot) {
      if (callerAsync && ownStream && hostCopyStream) {
        handOverToHost(rows, out, channels, start, frames);   // (leaves on the copy stream: ga_engine.hpp)
      } else {
        for (int ch = 0; ch < channels; ch++)
          GA_HIP(hipMemcpyAsync(out[ch] + start, rows[ch], (size_t)frames * sizeof(float), hipMemcpyDeviceToHost, stream));
      }
    }
  } catch (...) {
    asyncMode = callerAsync;
    throw;
  }
  asyncMode = callerAsync;
  if (!callerAsync) {
    GA_HIP(hipStreamSynchronize(stream));
    harvestProfile(true);
  }
}

}  // namespace ga
