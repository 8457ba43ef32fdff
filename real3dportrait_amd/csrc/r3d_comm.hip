// Frame gather over RCCL behind the C ABI (SURVEY 8(b) last row, 8(e)): frames of a driving clip are rendered frame-sharded, one
// process per GPU, and re-assembled on one rank.  There is no reduction and no ring: every peer has its own xGMI link to the root, so
// the collective is a gather written as grouped point-to-point ncclSend / ncclRecv (ncclGroupStart .. ncclGroupEnd) on the caller's
// stream.  The reference has no counterpart (its frame loop is serial, inference/real3d_infer.py:480-492).
//
// RCCL is resolved at run time (dlopen): a process that never calls r3d_comm_* does not need it, and a process that already has a
// copy mapped (torch.distributed ships its own librccl.so) reuses that one instead of loading a second RCCL.
#include <dlfcn.h>

#include "r3d_common.h"

namespace r3d {
namespace {

typedef struct { char internal[128]; } nccl_id;        // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128, rccl.h:40-43)
typedef void* nccl_comm;
enum { kNcclUint8 = 1 };                               // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1 (rccl.h)

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(nccl_id*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*CommCount)(nccl_comm, int*) = nullptr;
    int (*CommUserRank)(nccl_comm, int*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Rccl* rccl()
{
    static Rccl R;
    static bool tried = false;
    if (tried) return R.h ? &R : nullptr;
    tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (int pass = 0; pass < 2 && !R.h; ++pass)                    // pass 0: only a copy that is already mapped (torch's)
        for (const char* nm : names) {
            R.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (R.h) break;
        }
    if (!R.h) return nullptr;
#define R3D_SYM(field, sym) R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.h, sym)); if (!R.field) { R.h = nullptr; return nullptr; }
    R3D_SYM(GetUniqueId, "ncclGetUniqueId") R3D_SYM(CommInitRank, "ncclCommInitRank") R3D_SYM(CommDestroy, "ncclCommDestroy")
    R3D_SYM(CommCount, "ncclCommCount") R3D_SYM(CommUserRank, "ncclCommUserRank") R3D_SYM(GroupStart, "ncclGroupStart")
    R3D_SYM(GroupEnd, "ncclGroupEnd") R3D_SYM(Send, "ncclSend") R3D_SYM(Recv, "ncclRecv") R3D_SYM(GetErrorString, "ncclGetErrorString")
#undef R3D_SYM
    return &R;
}

int fail(Rccl* R, const char* what, int rc)
{
    set_error("%s: RCCL error %d (%s)", what, rc, R && R->GetErrorString ? R->GetErrorString(rc) : "?");
    return R3D_ERR_LAUNCH;
}

}  // namespace
}  // namespace r3d

using namespace r3d;

extern "C" int r3d_comm_unique_id(void* id128)
{
    Rccl* R = rccl();
    if (!R) { set_error("comm_unique_id: librccl.so not found"); return R3D_ERR_UNSUPPORTED; }
    if (!id128) { set_error("comm_unique_id: NULL"); return R3D_ERR_INVALID_ARG; }
    const int rc = R->GetUniqueId(reinterpret_cast<nccl_id*>(id128));
    return rc ? fail(R, "comm_unique_id", rc) : R3D_OK;
}

extern "C" int r3d_comm_init(const void* id128, int rank, int world, void** comm)
{
    Rccl* R = rccl();
    if (!R) { set_error("comm_init: librccl.so not found"); return R3D_ERR_UNSUPPORTED; }
    if (!id128 || !comm || world <= 0 || rank < 0 || rank >= world) { set_error("comm_init: bad argument"); return R3D_ERR_INVALID_ARG; }
    nccl_id id = *reinterpret_cast<const nccl_id*>(id128);
    nccl_comm c = nullptr;
    const int rc = R->CommInitRank(&c, world, id, rank);          // on the calling thread's current HIP device
    if (rc) return fail(R, "comm_init", rc);
    *comm = c;
    return R3D_OK;
}

extern "C" int r3d_comm_destroy(void* comm)
{
    Rccl* R = rccl();
    if (!R || !comm) return R3D_OK;
    const int rc = R->CommDestroy(comm);
    return rc ? fail(R, "comm_destroy", rc) : R3D_OK;
}

extern "C" int r3d_gather_frames(void* comm, const uint8_t* local, size_t bytes_per_rank, uint8_t* root_buf, int root, r3d_stream_t stream)
{
    Rccl* R = rccl();
    if (!R) { set_error("gather_frames: librccl.so not found"); return R3D_ERR_UNSUPPORTED; }
    if (!comm || !local || bytes_per_rank == 0) { set_error("gather_frames: bad argument"); return R3D_ERR_INVALID_ARG; }
    int world = 0, rank = -1, rc;
    if ((rc = R->CommCount(comm, &world)) || (rc = R->CommUserRank(comm, &rank))) return fail(R, "gather_frames", rc);
    if (root < 0 || root >= world || (rank == root && !root_buf)) { set_error("gather_frames: bad root / NULL root buffer"); return R3D_ERR_INVALID_ARG; }
    hipStream_t st = (hipStream_t)stream;
    if ((rc = R->GroupStart())) return fail(R, "gather_frames", rc);
    if (rank == root)
        for (int r = 0; r < world && !rc; ++r) rc = R->Recv(root_buf + (size_t)r * bytes_per_rank, bytes_per_rank, kNcclUint8, r, comm, st);
    if (!rc) rc = R->Send(local, bytes_per_rank, kNcclUint8, root, comm, st);
    const int rc2 = R->GroupEnd();
    if (rc || rc2) return fail(R, "gather_frames", rc ? rc : rc2);
    return R3D_OK;
}
