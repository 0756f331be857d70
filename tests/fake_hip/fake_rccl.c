/*
 * fake_rccl.c - TEST INFRASTRUCTURE: the seven RCCL entry points eddsa_amd.c looks up in "librccl.so.1"
 * (ncclCommInitAll, ncclCommDestroy, ncclAllGather, ncclBroadcast, ncclGroupStart, ncclGroupEnd,
 * ncclGetErrorString), for ONE process driving several fake devices.  Built as tests/fake_hip/_build/librccl.so.1 and
 * found through LD_LIBRARY_PATH by the test binaries only.
 *
 * Collectives are recorded between ncclGroupStart and ncclGroupEnd and executed at ncclGroupEnd, which is where the real
 * library would deadlock or corrupt memory if the single-process call pattern were wrong.  Checked, with a message and
 * ncclInvalidUsage / ncclInvalidArgument as the result:
 *   - every rank of the communicator takes part in every collective of the group, with the same count (and root);
 *   - rank r's buffers live on the device the communicator was created for, and so does its stream;
 *   - all-gather in place obeys the rule  sendbuff == recvbuff + rank * count  (any other overlap is an error).
 * Data movement is then what the collectives mean: memcpy between the "devices'" host memory.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fake_hip.h"

enum { ncclSuccess = 0, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
struct ncclComm { int rank, nranks, device; struct clique *cl; };
struct clique { int nranks; struct ncclComm *comm[64]; };
typedef struct ncclComm *ncclComm_t;

enum { OP_ALLGATHER, OP_BROADCAST };
struct op { int kind, root; const void *send; void *recv; size_t count; struct ncclComm *comm; hipStream_t stream; };
#define MAX_OPS 4096
static struct op g_ops[MAX_OPS];
static int g_nops, g_depth;
static pthread_mutex_t g_lk = PTHREAD_MUTEX_INITIALIZER;
static long g_gathers, g_broadcasts;                       /* collectives executed (read by the test through fake_rccl_stats) */

static int complain(int rc, const char *what)
{
    fprintf(stderr, "fake_rccl: %s\n", what);
    return rc;
}

/* fault injection for tests/c/host_fault_walk.c, as in fake_hip.c: the call with this ordinal (ncclCommInitAll, group
 * start / end and the collectives are counted) fails instead of doing its work; one shot */
static long g_calls, g_fail_at, g_fired;         /* atomics */
long fake_rccl_calls(void) { return __atomic_load_n(&g_calls, __ATOMIC_SEQ_CST); }
void fake_rccl_fail_call(long ordinal) { __atomic_store_n(&g_fail_at, ordinal, __ATOMIC_SEQ_CST); }
long fake_rccl_faults_fired(void) { return __atomic_load_n(&g_fired, __ATOMIC_SEQ_CST); }
static int tick(void)
{
    const long c = __atomic_add_fetch(&g_calls, 1, __ATOMIC_SEQ_CST);
    long at = c;
    if (!__atomic_compare_exchange_n(&g_fail_at, &at, 0, 0, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return 0;
    __atomic_add_fetch(&g_fired, 1, __ATOMIC_SEQ_CST);
    return 1;
}
#define ncclSystemError_ 2

int ncclCommInitAll(ncclComm_t *comms, int n, const int *devices)
{
    if (n < 1 || n > 64) return complain(ncclInvalidArgument, "ncclCommInitAll: bad rank count");
    if (tick()) return ncclSystemError_;
    struct clique *cl = (struct clique *)calloc(1, sizeof(*cl));
    cl->nranks = n;
    for (int r = 0; r < n; r++) {
        const int dev = devices ? devices[r] : r;
        if (dev < 0 || dev >= fake_hip_device_count()) return complain(ncclInvalidArgument, "ncclCommInitAll: no such device");
        for (int k = 0; k < r; k++) if (cl->comm[k]->device == dev) return complain(ncclInvalidUsage, "ncclCommInitAll: two ranks on one device");
        comms[r] = (struct ncclComm *)calloc(1, sizeof(struct ncclComm));
        comms[r]->rank = r; comms[r]->nranks = n; comms[r]->device = dev; comms[r]->cl = cl;
        cl->comm[r] = comms[r];
    }
    return ncclSuccess;
}

int ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    struct clique *cl = c->cl;
    int left = 0;
    cl->comm[c->rank] = NULL;
    for (int r = 0; r < cl->nranks; r++) left += cl->comm[r] != NULL;
    if (!left) free(cl);
    free(c);
    return ncclSuccess;
}

const char *ncclGetErrorString(int r)
{
    return r == ncclSuccess ? "no error" : r == ncclInvalidArgument ? "invalid argument (fake RCCL)" : r == ncclInvalidUsage ? "invalid usage (fake RCCL)" : "error (fake RCCL)";
}

static int push(struct op o, hipStream_t stream)
{
    if (!o.comm) return complain(ncclInvalidArgument, "NULL communicator");
    if (fake_hip_stream_device(stream) != o.comm->device) return complain(ncclInvalidUsage, "the stream belongs to another device than the communicator's rank");
    if (o.count) {
        if (fake_hip_owner(o.recv, o.kind == OP_ALLGATHER ? o.count * (size_t)o.comm->nranks : o.count) != o.comm->device)
            return complain(ncclInvalidArgument, "receive buffer is not (entirely) memory of the rank's device");
        if (fake_hip_owner(o.send, o.count) != o.comm->device) return complain(ncclInvalidArgument, "send buffer is not memory of the rank's device");
    }
    o.stream = fake_hip_resolve_stream(stream);     /* (a NULL handle: the default stream of the device current NOW) */
    pthread_mutex_lock(&g_lk);
    if (g_nops == MAX_OPS) { pthread_mutex_unlock(&g_lk); return complain(ncclInvalidUsage, "too many grouped operations"); }
    g_ops[g_nops++] = o;
    const int grouped = g_depth > 0;
    pthread_mutex_unlock(&g_lk);
    if (!grouped) return complain(ncclInvalidUsage, "a collective of a single-process multi-rank communicator outside ncclGroupStart/End would block for ever");
    return ncclSuccess;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, ncclComm_t comm, hipStream_t stream)
{
    if (dtype != 1) return complain(ncclInvalidArgument, "only ncclUint8 is expected here");
    if (comm && count) {                          /* in place: exactly at the rank's own slice; any other overlap is an error */
        const uint8_t *s = (const uint8_t *)send, *r = (const uint8_t *)recv;
        const size_t total = count * (size_t)comm->nranks;
        if (s + count > r && s < r + total && s != r + (size_t)comm->rank * count)
            return complain(ncclInvalidArgument, "in-place all-gather: sendbuff must be recvbuff + rank * count");
    }
    struct op o = { OP_ALLGATHER, 0, send, recv, count, comm, NULL };
    if (tick()) return ncclSystemError_;
    return push(o, stream);
}

int ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, ncclComm_t comm, hipStream_t stream)
{
    if (dtype != 1) return complain(ncclInvalidArgument, "only ncclUint8 is expected here");
    if (comm && (root < 0 || root >= comm->nranks)) return complain(ncclInvalidArgument, "broadcast root out of range");
    struct op o = { OP_BROADCAST, root, send, recv, count, comm, NULL };
    if (tick()) return ncclSystemError_;
    return push(o, stream);
}

int ncclGroupStart(void)
{
    if (tick()) return ncclSystemError_;
    pthread_mutex_lock(&g_lk);
    g_depth++;
    pthread_mutex_unlock(&g_lk);
    return ncclSuccess;
}

struct coll {
    int kind, root, nranks, left, leader;
    volatile int arrived[64];
    size_t count;
    const void *send[64]; void *recv[64]; hipStream_t stream[64];
};
struct coll_part { struct coll *c; int rank; };
static void coll_move(struct coll *c)
{
    const size_t count = c->count;
    if (c->kind == OP_ALLGATHER) {
        uint8_t *all = (uint8_t *)malloc(count * (size_t)c->nranks);    /* every contribution first: in-place receivers overwrite their senders */
        for (int r = 0; r < c->nranks; r++) memcpy(all + (size_t)r * count, c->send[r], count);
        for (int r = 0; r < c->nranks; r++) memcpy(c->recv[r], all, count * (size_t)c->nranks);
        free(all);
    } else {
        uint8_t *val = (uint8_t *)malloc(count);
        memcpy(val, c->send[c->root], count);
        for (int r = 0; r < c->nranks; r++) memcpy(c->recv[r], val, count);
        free(val);
    }
}
/* rank r's stream has reached the collective (a task of that stream; the fake runtime runs one task at a time) */
static void coll_arrive(void *a)
{
    struct coll_part *cp = (struct coll_part *)a;
    struct coll *c = cp->c;
    const int r = cp->rank;
    free(cp);
    c->arrived[r] = 1;
    if (c->leader < 0) {                         /* the first to arrive brings the others here, then moves the data */
        c->leader = r;
        for (int k = 0; k < c->nranks; k++) if (!c->arrived[k]) fake_hip_run_until(c->stream[k], &c->arrived[k]);
        coll_move(c);
    }
    if (--c->left == 0) free(c);
}

/* the k-th operation issued on every communicator of a clique forms one collective */
int ncclGroupEnd(void)
{
    int rc = ncclSuccess;
    pthread_mutex_lock(&g_lk);
    if (tick()) { g_depth = 0; g_nops = 0; pthread_mutex_unlock(&g_lk); return ncclSystemError_; }   /* (the group is abandoned) */
    if (g_depth == 0) { pthread_mutex_unlock(&g_lk); return complain(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart"); }
    if (--g_depth > 0) { pthread_mutex_unlock(&g_lk); return ncclSuccess; }
    int used[MAX_OPS] = { 0 };
    for (int i = 0; i < g_nops && rc == ncclSuccess; i++) {
        if (used[i]) continue;
        struct clique *cl = g_ops[i].comm->cl;
        struct op *part[64] = { 0 };
        /* the first not yet used operation of every rank of this clique */
        for (int r = 0; r < cl->nranks; r++)
            for (int k = i; k < g_nops; k++)
                if (!used[k] && g_ops[k].comm->cl == cl && g_ops[k].comm->rank == r) { part[r] = &g_ops[k]; used[k] = 1; break; }
        for (int r = 0; r < cl->nranks && rc == ncclSuccess; r++) {
            if (!part[r]) rc = g_fired ? ncclInvalidUsage : complain(ncclInvalidUsage, "a rank did not take part in a collective of the group (the real library would hang)");
            else if (part[r]->kind != part[0]->kind || part[r]->count != part[0]->count || part[r]->root != part[0]->root)
                rc = g_fired ? ncclInvalidUsage : complain(ncclInvalidUsage, "the ranks disagree on a collective's kind, count or root");   /* (expected once a call of the group was failed on purpose) */
        }
        if (rc != ncclSuccess) break;
        const size_t count = part[0]->count;
        if (!count) continue;
        /* the collective is work of every rank's stream: it moves the data when each of them has reached it (eager
         * model: now; FAKE_HIP_DEFER=1: when the first of the streams is made to run that far - which then makes the
         * others run up to their part) */
        struct coll *c = (struct coll *)calloc(1, sizeof(*c));
        c->kind = part[0]->kind; c->root = part[0]->root; c->count = count; c->nranks = cl->nranks; c->left = cl->nranks; c->leader = -1;
        for (int r = 0; r < cl->nranks; r++) { c->send[r] = part[r]->send; c->recv[r] = part[r]->recv; c->stream[r] = part[r]->stream; }
        if (c->kind == OP_ALLGATHER) g_gathers++; else g_broadcasts++;
        if (!fake_hip_deferred()) { coll_move(c); free(c); continue; }
        void *parts[64];
        for (int r = 0; r < cl->nranks; r++) {
            struct coll_part *cp = (struct coll_part *)malloc(sizeof(*cp));
            cp->c = c; cp->rank = r;
            parts[r] = cp;
        }
        fake_hip_enqueue_group(cl->nranks, c->stream, coll_arrive, parts);
    }
    g_nops = 0;
    pthread_mutex_unlock(&g_lk);
    return rc;
}

void fake_rccl_stats(long out[2]) { out[0] = g_gathers; out[1] = g_broadcasts; }
