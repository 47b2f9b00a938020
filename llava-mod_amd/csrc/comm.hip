// C-ABI collectives of the data-parallel step over RCCL (`*_allreduce_grads`, `*_moe_all_to_all`).
//
// What they replace: the gradient averaging DeepSpeed's engine performs for the reference at every optimizer boundary
// (train/align_trainer.py:326-434 builds the engine from config/dpconfig/zero2*.json; ZeRO-2 = reduce-scatter of the gradients +
// all-gather of the updated parameters) and the two all-to-alls of deepspeed.moe.sharded_moe.MOELayer.forward around the expert
// FFNs (call site llava_qwen2_moe.py:536-546).  One communicator per process (= per GPU); every call is enqueued on the caller's
// HIP stream and returns at once.  xGMI is point-to-point, so the expert exchange is a grouped ncclSend/ncclRecv with one
// message per peer carrying ONLY the live rows (unequal counts), not a padded equal-split all-to-all.
//
// RCCL is bound at RUN time (dlopen; an instance already loaded by the host process — e.g. PyTorch's — is reused): the kernel
// library has no link-time dependency on it and single-GPU use never touches it.  The Python package drives these exchanges
// through torch.distributed (the same RCCL); these entry points are the boundary for a host that does not.
#include "common.h"
#include <dlfcn.h>
#include <stddef.h>

namespace {
typedef struct { char internal[128]; } uid_t128;       // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* comm_t;
enum { kSum = 0, kF32 = 7, kBF16 = 9 };                 // ncclSum, ncclFloat32, ncclBfloat16 (rccl.h)
struct Api {
  int (*GetUniqueId)(uid_t128*);
  int (*CommInitRank)(comm_t*, int, uid_t128, int);
  int (*CommDestroy)(comm_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
  int (*ReduceScatter)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
  int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t);
  int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t);
  int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t);
  int (*GroupStart)();
  int (*GroupEnd)();
  bool ok;
};
Api* api() {
  static Api a = [] {
    Api x{};
    void* h = nullptr;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // reuse the host's instance if there is one
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) return x;
#define LMOD_SYM(field, sym) *(void**)(&x.field) = dlsym(h, sym); if (!x.field) return x;
    LMOD_SYM(GetUniqueId, "ncclGetUniqueId") LMOD_SYM(CommInitRank, "ncclCommInitRank") LMOD_SYM(CommDestroy, "ncclCommDestroy")
    LMOD_SYM(AllReduce, "ncclAllReduce") LMOD_SYM(ReduceScatter, "ncclReduceScatter") LMOD_SYM(AllGather, "ncclAllGather")
    LMOD_SYM(Send, "ncclSend") LMOD_SYM(Recv, "ncclRecv") LMOD_SYM(GroupStart, "ncclGroupStart") LMOD_SYM(GroupEnd, "ncclGroupEnd")
#undef LMOD_SYM
    x.ok = true;
    return x;
  }();
  return &a;
}
struct Comm { comm_t c; int rank, world; };
inline int dt(int dtype) { return dtype == 0 ? kF32 : dtype == 1 ? kBF16 : -1; }
}  // namespace

extern "C" {

// rank 0 creates the 128-byte id and hands it to the other ranks by whatever channel the host has (file, socket, MPI, ...)
int lmod_comm_unique_id(void* id128) {
  if (!id128) return LMOD_EINVAL;
  if (!api()->ok) return LMOD_EUNSUPPORTED;
  return api()->GetUniqueId((uid_t128*)id128) == 0 ? LMOD_OK : LMOD_ELAUNCH;
}

// collective over all `world` ranks; the calling thread's current HIP device is the rank's GPU
int lmod_comm_init(void** comm, const void* id128, int rank, int world) {
  if (!comm || !id128 || world <= 0 || rank < 0 || rank >= world) return LMOD_EINVAL;
  if (!api()->ok) return LMOD_EUNSUPPORTED;
  uid_t128 id;
  __builtin_memcpy(&id, id128, sizeof(id));
  comm_t c = nullptr;
  if (api()->CommInitRank(&c, world, id, rank) != 0) return LMOD_ELAUNCH;
  *comm = new Comm{c, rank, world};
  return LMOD_OK;
}

int lmod_comm_destroy(void* comm) {
  if (!comm) return LMOD_EINVAL;
  Comm* m = (Comm*)comm;
  const int rc = api()->CommDestroy(m->c);
  delete m;
  return rc == 0 ? LMOD_OK : LMOD_ELAUNCH;
}

// buf[0..n) <- SUM over ranks, in place (dtype 0: fp32, 1: bf16).  The mean's 1/world is folded into lmod_adamw_step's grad_scale.
int lmod_allreduce_grads(void* comm, void* buf, long long n, int dtype, hipStream_t stream) {
  if (!comm || n < 0 || dt(dtype) < 0) return LMOD_EINVAL;
  if (n == 0) return LMOD_OK;
  if (!buf) return LMOD_EINVAL;
  Comm* m = (Comm*)comm;
  return api()->AllReduce(buf, buf, (size_t)n, dt(dtype), kSum, m->c, stream) == 0 ? LMOD_OK : LMOD_ELAUNCH;
}

// ZeRO-2 gradient phase: span[0 .. world*n_per_rank) -> chunk `rank` of the SUM, written IN PLACE at span + rank*n_per_rank
int lmod_reduce_scatter_grads(void* comm, void* span, long long n_per_rank, int dtype, hipStream_t stream) {
  if (!comm || n_per_rank < 0 || dt(dtype) < 0) return LMOD_EINVAL;
  if (n_per_rank == 0) return LMOD_OK;
  if (!span) return LMOD_EINVAL;
  Comm* m = (Comm*)comm;
  char* mine = (char*)span + (size_t)m->rank * (size_t)n_per_rank * (dtype == 0 ? 4 : 2);
  return api()->ReduceScatter(span, mine, (size_t)n_per_rank, dt(dtype), kSum, m->c, stream) == 0 ? LMOD_OK : LMOD_ELAUNCH;
}

// ZeRO-2 parameter phase: every rank publishes its updated chunk (at span + rank*n_per_rank), in place
int lmod_allgather_params(void* comm, void* span, long long n_per_rank, int dtype, hipStream_t stream) {
  if (!comm || n_per_rank < 0 || dt(dtype) < 0) return LMOD_EINVAL;
  if (n_per_rank == 0) return LMOD_OK;
  if (!span) return LMOD_EINVAL;
  Comm* m = (Comm*)comm;
  const char* mine = (const char*)span + (size_t)m->rank * (size_t)n_per_rank * (dtype == 0 ? 4 : 2);
  return api()->AllGather(mine, span, (size_t)n_per_rank, dt(dtype), m->c, stream) == 0 ? LMOD_OK : LMOD_ELAUNCH;
}

// Expert-parallel exchange of PACKED live rows (bf16, H wide): send_rows[d] consecutive rows of `send` go to peer d,
// recv_rows[s] rows from peer s land consecutively in `recv` (host arrays of `world` counts each — the ranks exchange their
// per-expert live counts first, cf. MoE._forward_expert_parallel).  The backward exchange is the same call with the roles swapped.
int lmod_moe_all_to_all(void* comm, const void* send, void* recv, const long long* send_rows, const long long* recv_rows, int H,
                        hipStream_t stream) {
  if (!comm || !send_rows || !recv_rows || H <= 0) return LMOD_EINVAL;
  Comm* m = (Comm*)comm;
  long long st = 0, rt = 0;
  for (int r = 0; r < m->world; ++r) { if (send_rows[r] < 0 || recv_rows[r] < 0) return LMOD_EINVAL; st += send_rows[r]; rt += recv_rows[r]; }
  if ((st > 0 && !send) || (rt > 0 && !recv)) return LMOD_EINVAL;
  if (api()->GroupStart() != 0) return LMOD_ELAUNCH;
  size_t so = 0, ro = 0;
  int bad = 0;
  for (int r = 0; r < m->world; ++r) {
    const size_t sn = (size_t)send_rows[r] * (size_t)H, rn = (size_t)recv_rows[r] * (size_t)H;
    if (sn) bad |= api()->Send((const char*)send + so * 2, sn, kBF16, r, m->c, stream);
    if (rn) bad |= api()->Recv((char*)recv + ro * 2, rn, kBF16, r, m->c, stream);
    so += sn; ro += rn;
  }
  bad |= api()->GroupEnd();
  return bad == 0 ? LMOD_OK : LMOD_ELAUNCH;
}

}  // extern "C"
