// TEST DOUBLE for RCCL on a ONE-GPU box (tests/test_rccl_stub.py): LD_PRELOADed in front of librccl.so, it serves the
// twelve entry points libryujin_hip.so imports -- ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
// ncclComm{Count,UserRank,CuDevice}, ncclGroupStart/End, ncclSend, ncclRecv, ncclAllReduce, ncclGetErrorString -- for
// several PROCESSES that share one device (which RCCL itself refuses), so that the library's multi-rank branch
// (`!comm->local`: ryujin_hip.hip, exchange_vector / exchange_matrix_rows / all-reduces; counts, peers, offsets,
// issue order, the two-stream choreography) executes as real processes before the first multi-GPU box does it over
// xGMI. NOT a transport anyone should ship: messages are staged through a POSIX shared-memory segment.
//
// Everything is STREAM ORDERED, as RCCL's calls are, and nothing blocks a host thread:
//   send   : hipStreamWaitValue64(slot free) ; hipMemcpyAsync(device -> shared slot) ; hipStreamWriteValue64(head)
//   recv   : hipStreamWaitValue64(head)      ; hipMemcpyAsync(shared slot -> device) ; hipStreamWriteValue64(tail)
//   reduce : every rank publishes its operand the same way, waits for the others', reduces IN RANK ORDER in a host
//            function (data complete by then: it never waits) and copies the result back.
// One channel of kSlots slots per ordered pair of ranks; a group (ncclGroupStart/End) enqueues all of its sends before
// its receives.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace
{
  constexpr int kMaxRanks = 8;
  constexpr int kSlots = 4;
  constexpr size_t kSlotBytes = 1u << 20; /* the test meshes exchange at most a few hundred KB per message */
  constexpr size_t kReduceCount = 64;

  struct Channel {
    volatile uint64_t head; /* messages published by the sender */
    volatile uint64_t tail; /* messages consumed by the receiver */
    char pad[48];
  };

  struct Shared {
    std::atomic<int> attached;
    int n_ranks;
    char pad0[56];
    Channel channel[kMaxRanks][kMaxRanks]; /* [src][dst] */
    volatile uint64_t reduce_published[2][kMaxRanks][8];
    volatile uint64_t reduce_done[2][kMaxRanks][8];
    double reduce_operand[2][kMaxRanks][kReduceCount];
    double reduce_result[kMaxRanks][2][kReduceCount];
    /* slots follow: [src][dst][kSlots][kSlotBytes] */
  };

  size_t segment_bytes() { return sizeof(Shared) + (size_t)kMaxRanks * kMaxRanks * kSlots * kSlotBytes; }

  struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    hipStream_t stream;
  };
}

struct ncclComm {
  Shared *shared = nullptr;
  char *slots = nullptr;
  int rank = 0, n_ranks = 0, device = 0;
  uint64_t sent[kMaxRanks] = {}, received[kMaxRanks] = {};
  uint64_t reductions = 0;
  std::string name;
};

namespace
{
  thread_local int g_group_depth = 0;
  thread_local std::vector<std::pair<ncclComm *, Op>> g_group;

  char *slot_of(ncclComm *c, int src, int dst, uint64_t seq)
  {
    return c->slots + ((((size_t)src * kMaxRanks + dst) * kSlots + seq % kSlots) * kSlotBytes);
  }

  bool ok(hipError_t e) { return e == hipSuccess; }

  ncclResult_t issue(ncclComm *c, const Op &op)
  {
    if (op.bytes > kSlotBytes || op.peer < 0 || op.peer >= c->n_ranks || op.peer == c->rank)
      return ncclInvalidArgument;
    if (op.bytes == 0)
      return ncclSuccess;
    if (op.send) {
      Channel &ch = c->shared->channel[c->rank][op.peer];
      const uint64_t seq = c->sent[op.peer]++;
      if (seq >= (uint64_t)kSlots &&
          !ok(hipStreamWaitValue64(op.stream, (void *)&ch.tail, seq + 1 - kSlots, hipStreamWaitValueGte, ~0ull)))
        return ncclUnhandledCudaError;
      if (!ok(hipMemcpyAsync(slot_of(c, c->rank, op.peer, seq), op.buf, op.bytes, hipMemcpyDeviceToHost, op.stream)))
        return ncclUnhandledCudaError;
      if (!ok(hipStreamWriteValue64(op.stream, (void *)&ch.head, seq + 1, 0)))
        return ncclUnhandledCudaError;
    } else {
      Channel &ch = c->shared->channel[op.peer][c->rank];
      const uint64_t seq = c->received[op.peer]++;
      if (!ok(hipStreamWaitValue64(op.stream, (void *)&ch.head, seq + 1, hipStreamWaitValueGte, ~0ull)))
        return ncclUnhandledCudaError;
      if (!ok(hipMemcpyAsync(op.buf, slot_of(c, op.peer, c->rank, seq), op.bytes, hipMemcpyHostToDevice, op.stream)))
        return ncclUnhandledCudaError;
      if (!ok(hipStreamWriteValue64(op.stream, (void *)&ch.tail, seq + 1, 0)))
        return ncclUnhandledCudaError;
    }
    return ncclSuccess;
  }

  size_t size_of(ncclDataType_t t)
  {
    switch (t) {
    case ncclDouble:
      return 8;
    case ncclInt:
      return 4;
    default:
      return 0;
    }
  }

  struct ReduceJob {
    ncclComm *comm;
    int parity;
    size_t count;
    ncclDataType_t type;
    ncclRedOp_t op;
  };

  void reduce_on_host(void *p)
  {
    ReduceJob *job = static_cast<ReduceJob *>(p);
    ncclComm *c = job->comm;
    double *out = c->shared->reduce_result[c->rank][job->parity];
    for (size_t q = 0; q < job->count; ++q) {
      if (job->type == ncclDouble) {
        double acc = c->shared->reduce_operand[job->parity][0][q];
        for (int r = 1; r < c->n_ranks; ++r) { /* rank order: the same bits on every rank */
          const double v = c->shared->reduce_operand[job->parity][r][q];
          acc = job->op == ncclMin ? (v < acc ? v : acc) : job->op == ncclMax ? (v > acc ? v : acc) : acc + v;
        }
        out[q] = acc;
      } else {
        auto operand = [&](int r) { return reinterpret_cast<const int *>(c->shared->reduce_operand[job->parity][r])[q]; };
        int acc = operand(0);
        for (int r = 1; r < c->n_ranks; ++r) {
          const int v = operand(r);
          acc = job->op == ncclMin ? (v < acc ? v : acc) : job->op == ncclMax ? (v > acc ? v : acc) : acc + v;
        }
        reinterpret_cast<int *>(out)[q] = acc;
      }
    }
    delete job;
  }
}

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "/ryujin-rccl-stub-%d-%lx", (int)getpid(), (unsigned long)random());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int n_ranks, ncclUniqueId id, int rank)
{
  if (n_ranks < 1 || n_ranks > kMaxRanks || rank < 0 || rank >= n_ranks)
    return ncclInvalidArgument;
  auto *c = new ncclComm;
  c->rank = rank;
  c->n_ranks = n_ranks;
  c->name = id.internal;
  if (!ok(hipGetDevice(&c->device)))
    return ncclUnhandledCudaError;
  const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)segment_bytes()) != 0)
    return ncclSystemError;
  void *base = mmap(nullptr, segment_bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (base == MAP_FAILED)
    return ncclSystemError;
  /* the flags the streams wait on / write and the staging slots: visible to the device */
  if (!ok(hipHostRegister(base, segment_bytes(), hipHostRegisterPortable | hipHostRegisterMapped)))
    return ncclUnhandledCudaError;
  c->shared = static_cast<Shared *>(base);
  c->slots = static_cast<char *>(base) + sizeof(Shared);
  c->shared->n_ranks = n_ranks;
  /* (a fresh segment is zero filled: every counter starts at 0) -- wait until all ranks are attached, as
   * ncclCommInitRank does */
  c->shared->attached.fetch_add(1);
  while (c->shared->attached.load() < n_ranks)
    usleep(1000);
  if (rank == 0)
    shm_unlink(c->name.c_str()); /* everybody has it mapped: nothing is left behind if a rank dies */
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
  if (!c)
    return ncclSuccess;
  (void)hipDeviceSynchronize();
  (void)hipHostUnregister(c->shared);
  munmap(c->shared, segment_bytes());
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *count)
{
  *count = c->n_ranks;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t c, int *rank)
{
  *rank = c->rank;
  return ncclSuccess;
}

ncclResult_t ncclCommCuDevice(const ncclComm_t c, int *device)
{
  *device = c->device;
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
  switch (r) {
  case ncclSuccess:
    return "no error";
  case ncclInvalidArgument:
    return "rccl stub: invalid argument (peer, datatype, or a message larger than a slot)";
  case ncclSystemError:
    return "rccl stub: shared memory segment";
  default:
    return "rccl stub: HIP call failed";
  }
}

ncclResult_t ncclGroupStart()
{
  ++g_group_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
  if (--g_group_depth > 0)
    return ncclSuccess;
  ncclResult_t res = ncclSuccess;
  for (const bool sends : {true, false})
    for (auto &it : g_group)
      if (it.second.send == sends && res == ncclSuccess)
        res = issue(it.first, it.second);
  g_group.clear();
  return res;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream)
{
  if (size_of(type) == 0)
    return ncclInvalidArgument;
  const Op op{true, const_cast<void *>(buf), count * size_of(type), peer, stream};
  if (g_group_depth > 0) {
    g_group.emplace_back(c, op);
    return ncclSuccess;
  }
  return issue(c, op);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream)
{
  if (size_of(type) == 0)
    return ncclInvalidArgument;
  const Op op{false, buf, count * size_of(type), peer, stream};
  if (g_group_depth > 0) {
    g_group.emplace_back(c, op);
    return ncclSuccess;
  }
  return issue(c, op);
}

ncclResult_t ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t type, ncclRedOp_t op,
                           ncclComm_t c, hipStream_t stream)
{
  const size_t bytes = count * size_of(type);
  if (size_of(type) == 0 || bytes > kReduceCount * sizeof(double) || !(op == ncclMin || op == ncclMax || op == ncclSum))
    return ncclInvalidArgument;
  Shared *s = c->shared;
  const uint64_t seq = c->reductions++;
  const int parity = (int)(seq & 1);
  /* the operand slot of this parity is free once every rank has finished reduction seq - 2 */
  if (seq >= 2)
    for (int r = 0; r < c->n_ranks; ++r)
      if (!ok(hipStreamWaitValue64(stream, (void *)&s->reduce_done[parity][r][0], seq - 1, hipStreamWaitValueGte, ~0ull)))
        return ncclUnhandledCudaError;
  if (!ok(hipMemcpyAsync(s->reduce_operand[parity][c->rank], sendbuf, bytes, hipMemcpyDeviceToHost, stream)) ||
      !ok(hipStreamWriteValue64(stream, (void *)&s->reduce_published[parity][c->rank][0], seq + 1, 0)))
    return ncclUnhandledCudaError;
  for (int r = 0; r < c->n_ranks; ++r)
    if (!ok(hipStreamWaitValue64(stream, (void *)&s->reduce_published[parity][r][0], seq + 1, hipStreamWaitValueGte, ~0ull)))
      return ncclUnhandledCudaError;
  if (!ok(hipLaunchHostFunc(stream, reduce_on_host, new ReduceJob{c, parity, count, type, op})) ||
      !ok(hipMemcpyAsync(recvbuf, s->reduce_result[c->rank][parity], bytes, hipMemcpyHostToDevice, stream)) ||
      !ok(hipStreamWriteValue64(stream, (void *)&s->reduce_done[parity][c->rank][0], seq + 1, 0)))
    return ncclUnhandledCudaError;
  return ncclSuccess;
}
}
