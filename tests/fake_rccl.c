/* fake_rccl.c -- TEST INFRASTRUCTURE, not product: a stand-in for librccl.so that moves data between PROCESSES ON ONE GPU through POSIX
 * shared memory, so that the multi-rank code of csrc/ugsm_shard.cpp -- the receiving side of ugsm_submit_fovea_shard, the per-slot state
 * buffers on a rank that never runs the coarse phase, ugsm_shard_gather's send / receive addressing, ugsm_shard_count_ranks -- can run with
 * two ranks on the one-GPU pool (RCCL itself refuses two ranks on one device).  tests/test_gpu_dist.py builds it
 * (gcc -shared -fPIC ... -lamdhip64) and points the library at it with UGSM_RCCL_PATH.
 *
 * It implements the dozen entry points ugsm_shard.cpp resolves, with RCCL's signatures (/opt/rocm/include/rccl/rccl.h), SYNCHRONOUSLY:
 * a collective waits for the caller's stream, copies through the shared segment and returns when the data is in place, so everything
 * enqueued on the stream afterwards sees it.  What this does NOT exercise is RCCL: transport, asynchrony, xGMI.  The protocol is
 * lock-step by construction -- every rank makes the same sequence of collective calls -- exactly RCCL's own rule. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclHalf = 6,
               ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;

#define MAX_RANKS 8
#define DATA_BYTES ((size_t)96 << 20) /* per rank */

typedef struct {
    volatile int barrier_count, barrier_gen;
    volatile long long send_seq[MAX_RANKS], ack_seq[MAX_RANKS];
    volatile float red[MAX_RANKS];
} Header;

struct ncclComm {
    int rank, world;
    Header *h;
    char *data; /* world x DATA_BYTES behind the header */
    size_t map_bytes;
    long long sent, got[MAX_RANKS];
};
typedef struct ncclComm *ncclComm_t;

static void nap(void)
{
    struct timespec ts = {0, 50000};
    nanosleep(&ts, NULL);
}

static int barrier(ncclComm_t c)
{
    const int gen = c->h->barrier_gen;
    if (__sync_add_and_fetch(&c->h->barrier_count, 1) == c->world) {
        c->h->barrier_count = 0;
        __sync_synchronize();
        c->h->barrier_gen = gen + 1;
        return 0;
    }
    for (long spins = 0; c->h->barrier_gen == gen; spins++) {
        nap();
        if (spins > 1200000) return -1; /* a minute: the peer is gone */
    }
    return 0;
}

static size_t dsize(ncclDataType_t t) { return t <= 1 ? 1 : (t == 6 ? 2 : (t == 4 || t == 5 || t == 8 ? 8 : 4)); }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/ugsm_fake_rccl_%d_%ld", (int)getpid(), (long)time(NULL));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const size_t bytes = 4096 + (size_t)nranks * DATA_BYTES;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return ncclSystemError;
    } else {
        for (int tries = 0; tries < 600 && fd < 0; tries++) { /* rank 0 may not have created it yet */
            fd = shm_open(id.internal, O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && (fstat(fd, &sb) != 0 || (size_t)sb.st_size < bytes)) {
                close(fd);
                fd = -1;
            }
            if (fd < 0) usleep(100000);
        }
        if (fd < 0) return ncclSystemError;
    }
    void *p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    ncclComm_t c = (ncclComm_t)calloc(1, sizeof *c);
    c->rank = rank;
    c->world = nranks;
    c->h = (Header *)p;
    c->data = (char *)p + 4096;
    c->map_bytes = bytes;
    *comm = c;
    if (barrier(c) != 0) return ncclSystemError; /* collective, like the real one */
    if (rank == 0) shm_unlink(id.internal);     /* everyone has it mapped */
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist)
{
    (void)comm; (void)ndev; (void)devlist;
    return ncclInvalidUsage; /* one process per rank only */
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    munmap((void *)c->h, c->map_bytes);
    free(c);
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t c) { return ncclCommDestroy(c); } /* (nothing of this stand-in ever runs on the device) */

ncclResult_t ncclCommCount(const ncclComm_t c, int *count)
{
    *count = c->world;
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t stream)
{
    const size_t bytes = count * dsize(t);
    if (bytes > DATA_BYTES || root < 0 || root >= c->world) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError; /* what the stream wrote (root) / still reads (others) is done */
    if (c->rank == root && hipMemcpy(c->data, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (barrier(c) != 0) return ncclSystemError;
    if (c->rank != root && hipMemcpy(recvbuff, c->data, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (c->rank == root && recvbuff != sendbuff && hipMemcpy(recvbuff, sendbuff, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return barrier(c) == 0 ? ncclSuccess : ncclSystemError; /* the root does not overwrite the segment before everyone has read it */
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t stream)
{
    if (count != 1 || t != ncclFloat || op != ncclSum) return ncclInvalidArgument; /* all ugsm_shard_count_ranks needs */
    float v = 0.0f;
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(&v, sendbuff, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    c->h->red[c->rank] = v;
    if (barrier(c) != 0) return ncclSystemError;
    float s = 0.0f;
    for (int r = 0; r < c->world; r++) s += c->h->red[r];
    if (hipMemcpy(recvbuff, &s, sizeof s, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return barrier(c) == 0 ? ncclSuccess : ncclSystemError;
}

/* point to point: the sender's region of the segment is a one-message mailbox (send_seq / ack_seq) */
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t stream)
{
    const size_t bytes = count * dsize(t);
    (void)peer;
    if (bytes > DATA_BYTES) return ncclInvalidArgument;
    for (long spins = 0; c->h->ack_seq[c->rank] != c->sent; spins++) { /* the previous message has been taken */
        nap();
        if (spins > 1200000) return ncclSystemError;
    }
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(c->data + (size_t)c->rank * DATA_BYTES, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    __sync_synchronize();
    c->h->send_seq[c->rank] = ++c->sent;
    return ncclSuccess;
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t stream)
{
    const size_t bytes = count * dsize(t);
    if (bytes > DATA_BYTES || peer < 0 || peer >= c->world) return ncclInvalidArgument;
    for (long spins = 0; c->h->send_seq[peer] <= c->got[peer]; spins++) {
        nap();
        if (spins > 1200000) return ncclSystemError;
    }
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(recvbuff, c->data + (size_t)peer * DATA_BYTES, bytes, hipMemcpyHostToDevice) != hipSuccess)
        return ncclUnhandledCudaError;
    c->got[peer]++;
    __sync_synchronize();
    c->h->ack_seq[peer] = c->got[peer];
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL error (tests/fake_rccl.c)"; }
