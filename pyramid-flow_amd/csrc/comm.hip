// Communicator of the C ABI (include/pyflow_hip.h, "multi-GPU"): RCCL over xGMI, one process per GPU.
//
// Replaces the collectives the reference issues through torch.distributed -- `dist.all_to_all` inside
// trainer_misc/communicate.py:7-26 (sequence parallelism), the isend / irecv halo pass and the list all_gather of
// video_vae/context_parallel_ops.py:41-114 (context parallelism), the per-step broadcast of
// pyramid_dit_for_video_gen_pipeline.py:752-756 -- for hosts that do not carry torch.distributed.  The Python host of
// this repository keeps torch.distributed (backend "nccl" = the same RCCL) as its default client and can be switched to
// this communicator (pyflow_hip/comm_native.py).
//
// Every collective is enqueued on the communicator's OWN HIP stream, ordered after the work already queued on the
// caller's compute stream (event), so that kernels launched next on the compute stream overlap with it; pf_comm_wait
// makes a stream wait for everything the communicator has queued (event, no host sync).  No allocation, no host
// synchronisation.  RCCL is resolved at run time (dlopen of the library the process already has: no link dependency,
// one RCCL per process even next to torch).
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "common.h"
#include "pyflow_hip.h"

int pf_set_err(const char* m);

namespace {

typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
enum { RCCL_INT8 = 0, RCCL_UINT8 = 1, RCCL_FLOAT32 = 7, RCCL_SUM = 0 };

struct Api {
    int (*GetUniqueId)(rcclUniqueId*);
    int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int);
    int (*CommDestroy)(rcclComm_t);
    int (*GroupStart)();
    int (*GroupEnd)();
    int (*Send)(const void*, size_t, int, int, rcclComm_t, hipStream_t);
    int (*Recv)(void*, size_t, int, int, rcclComm_t, hipStream_t);
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
    int (*Broadcast)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
    const char* (*GetErrorString)(int);
    bool ok = false;
};
Api g_api;

bool load_api() {
    if (g_api.ok) return true;
    void* h = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return false;
#define PF_SYM(field, name)                                  \
    *(void**)(&g_api.field) = dlsym(h, name);                \
    if (!g_api.field) return false;
    PF_SYM(GetUniqueId, "ncclGetUniqueId")
    PF_SYM(CommInitRank, "ncclCommInitRank")
    PF_SYM(CommDestroy, "ncclCommDestroy")
    PF_SYM(GroupStart, "ncclGroupStart")
    PF_SYM(GroupEnd, "ncclGroupEnd")
    PF_SYM(Send, "ncclSend")
    PF_SYM(Recv, "ncclRecv")
    PF_SYM(AllReduce, "ncclAllReduce")
    PF_SYM(Broadcast, "ncclBroadcast")
    PF_SYM(GetErrorString, "ncclGetErrorString")
#undef PF_SYM
    g_api.ok = true;
    return true;
}

int rccl_fail(const char* where, int rc) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: RCCL error %d (%s)", where, rc, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
    return pf_set_err(buf);
}

}  // namespace

// COPY-ENGINE TRANSPORT (round 6; pf_comm_create_window / pf_comm_attach_windows).  Round 5 measured on one GPU that an
// RCCL send / recv KERNEL beside the persistent GEMM hides only ~0.35 of its time (a gemm8p workgroup owns its CU whole: the
// communication kernel waits for CUs or delays the GEMM), while a copy-engine transfer hides 0.71-0.84
// (profiles/r05_comm_overlap_bench.log).  With windows attached, pf_all_to_all_v / pf_halo_send_recv / pf_all_gather_v move
// their chunks WITHOUT kernels: every rank owns one IPC-shared window = `world` slots of `slot` bytes (slot s receives from
// rank s) + two flag words per peer; rank r copies its chunk for p into p's slot r (hipMemcpyAsync device-to-device between
// IPC-mapped allocations: SDMA over xGMI), then writes the exchange's sequence number into p's data flag r
// (hipStreamWriteValue32); p waits for it (hipStreamWaitValue32, >=), copies the slot to where the caller wants it and
// acknowledges into r's ack flag p -- r's next exchange waits for that acknowledgement before it overwrites the slot.
// Everything is stream-ordered on the communicator's stream: no host synchronisation, no CU held while waiting.
// Chunks larger than a slot go through RCCL as before (when the communicator has one).  NOT YET RUN BETWEEN TWO GPUS: the
// builder's boxes have one; tests/test_sp_gpu.py runs the protocol between two processes that share a GPU.
struct PeerWin {
    bool on = false;
    char* win = nullptr;            // my window: [world][slot] bytes, then data_flag[world], ack_flag[world] (uint32)
    long long slot = 0;
    char* peer[64] = {};            // every rank's window as mapped here (peer[rank] = win)
    unsigned sent[64] = {}, rcvd[64] = {};      // per PAIR: chunks sent to / received from each peer so far (both sides of a pair
                                                // count the same exchanges, so the sequence numbers agree without any global epoch)
};

struct pf_comm {
    rcclComm_t comm;         // nullptr: a communicator without RCCL (pf_comm_init_local: window transport only)
    int rank, world;
    hipStream_t stream;      // the communicator's own stream
    hipEvent_t ev_in;        // compute stream -> comm stream ordering
    hipEvent_t ev_out;       // comm stream -> waiting stream ordering
    PeerWin pw;
};

static unsigned* flag_ptr(char* window, long long slot, int world, int which, int idx) {      // which: 0 = data, 1 = ack
    return (unsigned*)(window + (long long)world * slot) + which * world + idx;
}

#define PF_RCCL(call, where)                      \
    do {                                          \
        const int rc_ = (call);                   \
        if (rc_ != 0) return rccl_fail(where, rc_); \
    } while (0)

// A call between GroupStart and GroupEnd that fails must not leave the communicator inside an open group (the next
// collective would be queued into it and hang instead of reporting): close the group, then report the FIRST error.
#define PF_RCCL_IN_GROUP(call, where)             \
    do {                                          \
        const int rc_ = (call);                   \
        if (rc_ != 0) {                           \
            g_api.GroupEnd();                     \
            return rccl_fail(where, rc_);         \
        }                                         \
    } while (0)

static int order_after(pf_comm* c, hipStream_t compute) {
    if (hipEventRecord(c->ev_in, compute) != hipSuccess) return pf_set_err("pf_comm: hipEventRecord failed");
    if (hipStreamWaitEvent(c->stream, c->ev_in, 0) != hipSuccess) return pf_set_err("pf_comm: hipStreamWaitEvent failed");
    return 0;
}

extern "C" int pf_comm_unique_id(void* out128) {
    if (!out128) return pf_set_err("pf_comm_unique_id: null buffer");
    if (!load_api()) return pf_set_err("pf_comm: librccl.so not found");
    rcclUniqueId id;
    PF_RCCL(g_api.GetUniqueId(&id), "pf_comm_unique_id");
    memcpy(out128, id.internal, 128);
    return 0;
}

extern "C" int pf_comm_init(pf_comm** out, int rank, int world, const void* unique_id_128) {
    if (!out || !unique_id_128 || world < 1 || rank < 0 || rank >= world) return pf_set_err("pf_comm_init: bad arguments");
    if (!load_api()) return pf_set_err("pf_comm: librccl.so not found");
    pf_comm* c = new pf_comm();
    c->rank = rank;
    c->world = world;
    rcclUniqueId id;
    memcpy(id.internal, unique_id_128, 128);
    const int rc = g_api.CommInitRank(&c->comm, world, id, rank);
    if (rc != 0) { delete c; return rccl_fail("pf_comm_init", rc); }
    c->stream = nullptr;
    c->ev_in = c->ev_out = nullptr;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) {
        // give back whatever was created before the failure: the RCCL communicator, the stream, the first event
        if (c->ev_out) hipEventDestroy(c->ev_out);
        if (c->ev_in) hipEventDestroy(c->ev_in);
        if (c->stream) hipStreamDestroy(c->stream);
        g_api.CommDestroy(c->comm);
        delete c;
        return pf_set_err("pf_comm_init: stream / event creation failed");
    }
    *out = c;
    return 0;
}

static void drop_windows(pf_comm* c) {
    if (!c->pw.win) return;
    for (int p = 0; p < c->world; ++p)
        if (p != c->rank && c->pw.peer[p]) hipIpcCloseMemHandle(c->pw.peer[p]);
    hipFree(c->pw.win);
    c->pw = PeerWin();
}

extern "C" int pf_comm_destroy(pf_comm* c) {
    if (!c) return 0;
    hipStreamSynchronize(c->stream);
    drop_windows(c);
    if (c->comm) g_api.CommDestroy(c->comm);
    hipEventDestroy(c->ev_in);
    hipEventDestroy(c->ev_out);
    hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

// a communicator WITHOUT RCCL: its own stream and events only; collectives work once windows are attached (and then only
// for chunks that fit a slot).  For hosts / tests whose ranks cannot form an RCCL communicator (several ranks on one GPU).
extern "C" int pf_comm_init_local(pf_comm** out, int rank, int world) {
    if (!out || world < 1 || world > 64 || rank < 0 || rank >= world) return pf_set_err("pf_comm_init_local: bad arguments");
    pf_comm* c = new pf_comm();
    c->comm = nullptr;
    c->rank = rank;
    c->world = world;
    c->stream = nullptr;
    c->ev_in = c->ev_out = nullptr;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) {
        if (c->ev_out) hipEventDestroy(c->ev_out);
        if (c->ev_in) hipEventDestroy(c->ev_in);
        if (c->stream) hipStreamDestroy(c->stream);
        delete c;
        return pf_set_err("pf_comm_init_local: stream / event creation failed");
    }
    *out = c;
    return 0;
}

// step 1 of the window bootstrap: allocate this rank's window (world slots of slot_bytes + flags, zeroed) and return its
// 64-byte IPC handle; the host ships every rank's handle to every rank (any byte channel), then calls pf_comm_attach_windows
extern "C" int pf_comm_create_window(pf_comm* c, long long slot_bytes, void* handle_out64) {
    if (!c || !handle_out64 || slot_bytes <= 0 || (slot_bytes & 255) || c->world > 64) return pf_set_err("pf_comm_create_window: bad arguments (slot_bytes: a multiple of 256)");
    if (c->pw.win) return pf_set_err("pf_comm_create_window: this communicator has a window");
    const long long total = (long long)c->world * slot_bytes + 2ll * c->world * 4 + 256;
    void* w = nullptr;
    if (hipExtMallocWithFlags(&w, (size_t)total, hipDeviceMallocFinegrained) != hipSuccess || !w) {
        (void)hipGetLastError();
        if (hipMalloc(&w, (size_t)total) != hipSuccess) return pf_set_err("pf_comm_create_window: allocation failed");
    }
    if (hipMemset(w, 0, (size_t)total) != hipSuccess) { hipFree(w); return pf_set_err("pf_comm_create_window: memset failed"); }
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, w) != hipSuccess) { hipFree(w); return pf_set_err("pf_comm_create_window: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 ?)"); }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    memcpy(handle_out64, &h, 64);
    c->pw.win = (char*)w;
    c->pw.slot = slot_bytes;
    c->pw.peer[c->rank] = (char*)w;
    return 0;
}

// step 2: map every other rank's window (handles: world x 64 bytes, entry [rank] ignored) and switch the v-collectives of
// this communicator to the copy-engine transport for chunks of at most slot_bytes
extern "C" int pf_comm_attach_windows(pf_comm* c, const void* handles) {
    if (!c || !handles || !c->pw.win) return pf_set_err("pf_comm_attach_windows: create the window first");
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + 64ll * p, 64);
        void* ptr = nullptr;
        if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !ptr) {
            (void)hipGetLastError();
            for (int q = 0; q < p; ++q)
                if (q != c->rank && c->pw.peer[q]) { hipIpcCloseMemHandle(c->pw.peer[q]); c->pw.peer[q] = nullptr; }
            return pf_set_err("pf_comm_attach_windows: hipIpcOpenMemHandle failed");
        }
        c->pw.peer[p] = (char*)ptr;
    }
    c->pw.on = true;
    return 0;
}
// 0 = RCCL kernels, 1 = copy engines through attached windows
extern "C" int pf_comm_transport(const pf_comm* c) { return c && c->pw.on ? 1 : 0; }

// one exchange through the windows: send_* / recv_* as in pf_all_to_all_v.  The pair (me -> p) takes part when to[p] here and
// from[me] on p: all pairs for the all-to-all / all-gather, the neighbours for the halo pass.  Sequence numbers are per pair.
static int window_exchange(pf_comm* c, const void* send, const long long* send_bytes, const long long* send_offs, void* recv,
                           const long long* recv_bytes, const long long* recv_offs, const bool* to, const bool* from,
                           const char* where) {
    PeerWin& w = c->pw;
    const int me = c->rank, P = c->world;
#define PF_HIP(call)                                                                   \
    do {                                                                               \
        if ((call) != hipSuccess) { (void)hipGetLastError(); return pf_set_err(where); } \
    } while (0)
    for (int p = 0; p < P; ++p) {
        if (p == me || !to[p]) continue;
        // p has copied my previous chunk out of its slot `me` (its acknowledgement lands in MY window)
        const unsigned e = ++w.sent[p];
        PF_HIP(hipStreamWaitValue32(c->stream, flag_ptr(w.win, w.slot, P, 1, p), e - 1, hipStreamWaitValueGte, 0xffffffffu));
        if (send_bytes[p] > 0)
            PF_HIP(hipMemcpyAsync(w.peer[p] + (long long)me * w.slot, (const char*)send + send_offs[p], (size_t)send_bytes[p],
                                  hipMemcpyDeviceToDevice, c->stream));
        PF_HIP(hipStreamWriteValue32(c->stream, flag_ptr(w.peer[p], w.slot, P, 0, me), e, 0));
    }
    if (send_bytes[me] > 0)
        PF_HIP(hipMemcpyAsync((char*)recv + recv_offs[me], (const char*)send + send_offs[me], (size_t)send_bytes[me],
                              hipMemcpyDeviceToDevice, c->stream));
    for (int p = 0; p < P; ++p) {
        if (p == me || !from[p]) continue;
        const unsigned e = ++w.rcvd[p];
        PF_HIP(hipStreamWaitValue32(c->stream, flag_ptr(w.win, w.slot, P, 0, p), e, hipStreamWaitValueGte, 0xffffffffu));
        if (recv_bytes[p] > 0)
            PF_HIP(hipMemcpyAsync((char*)recv + recv_offs[p], w.win + (long long)p * w.slot, (size_t)recv_bytes[p],
                                  hipMemcpyDeviceToDevice, c->stream));
        PF_HIP(hipStreamWriteValue32(c->stream, flag_ptr(w.peer[p], w.slot, P, 1, me), e, 0));
    }
#undef PF_HIP
    return 0;
}
extern "C" int pf_comm_rank(const pf_comm* c) { return c ? c->rank : -1; }
extern "C" int pf_comm_world(const pf_comm* c) { return c ? c->world : -1; }

// all-to-all with per-peer byte counts / offsets (counts of 0 allowed): one grouped ncclSend / ncclRecv per peer
extern "C" int pf_all_to_all_v(pf_comm* c, const void* send, const long long* send_bytes, const long long* send_offs,
                               void* recv, const long long* recv_bytes, const long long* recv_offs, hipStream_t compute) {
    if (!c || !send_bytes || !send_offs || !recv_bytes || !recv_offs) return pf_set_err("pf_all_to_all_v: null argument");
    if (order_after(c, compute)) return -1;
    // the route is decided PER PAIR, on a number both ends of the pair know (what r sends to p is what p receives from r):
    // chunks of at most a window slot travel through the windows, larger ones through RCCL -- no rank can end up waiting on a
    // transport its peer did not take
    bool via_win_to[64], via_win_from[64], rest = false;
    for (int p = 0; p < c->world; ++p) {
        // (a direction that carries nothing is skipped altogether, as the RCCL path skips it: the two ends of a pair must
        //  count the same exchanges -- a send / recv pair is an all-to-all in which only one pair takes part)
        via_win_to[p] = c->pw.on && (p == c->rank || (send_bytes[p] > 0 && send_bytes[p] <= c->pw.slot));
        via_win_from[p] = c->pw.on && (p == c->rank || (recv_bytes[p] > 0 && recv_bytes[p] <= c->pw.slot));
        rest = rest || (!via_win_to[p] && send_bytes[p] > 0) || (!via_win_from[p] && recv_bytes[p] > 0);
    }
    if (c->pw.on && window_exchange(c, send, send_bytes, send_offs, recv, recv_bytes, recv_offs, via_win_to, via_win_from,
                                    "pf_all_to_all_v: window transport failed")) return -1;
    if (!rest) return 0;
    if (!c->comm) return pf_set_err("pf_all_to_all_v: a chunk exceeds the window slot and this communicator has no RCCL");
    PF_RCCL(g_api.GroupStart(), "pf_all_to_all_v");
    for (int p = 0; p < c->world; ++p) {
        if (send_bytes[p] > 0 && !via_win_to[p])
            PF_RCCL_IN_GROUP(g_api.Send((const char*)send + send_offs[p], (size_t)send_bytes[p], RCCL_UINT8, p, c->comm, c->stream), "pf_all_to_all_v");
        if (recv_bytes[p] > 0 && !via_win_from[p])
            PF_RCCL_IN_GROUP(g_api.Recv((char*)recv + recv_offs[p], (size_t)recv_bytes[p], RCCL_UINT8, p, c->comm, c->stream), "pf_all_to_all_v");
    }
    PF_RCCL(g_api.GroupEnd(), "pf_all_to_all_v");
    return 0;
}

// halo pass of the temporal context parallelism: `bytes` go to rank + 1, rank - 1's arrive in `recv`; no wrap-around
extern "C" int pf_halo_send_recv(pf_comm* c, const void* send, void* recv, long long bytes, hipStream_t compute) {
    if (!c || bytes < 0) return pf_set_err("pf_halo_send_recv: bad arguments");
    if (order_after(c, compute)) return -1;
    if (c->pw.on && bytes > 0 && bytes <= c->pw.slot) {
        long long sb[64], rb[64], zo[64];
        bool to[64], from[64];
        for (int p = 0; p < c->world; ++p) {
            to[p] = p == c->rank + 1;
            from[p] = p == c->rank - 1;
            sb[p] = to[p] ? bytes : 0;
            rb[p] = from[p] ? bytes : 0;
            zo[p] = 0;
        }
        return window_exchange(c, send, sb, zo, recv, rb, zo, to, from, "pf_halo_send_recv: window transport failed");
    }
    if (!c->comm) return pf_set_err("pf_halo_send_recv: the halo exceeds the window slot and this communicator has no RCCL");
    PF_RCCL(g_api.GroupStart(), "pf_halo_send_recv");
    if (c->rank + 1 < c->world && bytes > 0)
        PF_RCCL_IN_GROUP(g_api.Send(send, (size_t)bytes, RCCL_UINT8, c->rank + 1, c->comm, c->stream), "pf_halo_send_recv");
    if (c->rank > 0 && bytes > 0)
        PF_RCCL_IN_GROUP(g_api.Recv(recv, (size_t)bytes, RCCL_UINT8, c->rank - 1, c->comm, c->stream), "pf_halo_send_recv");
    PF_RCCL(g_api.GroupEnd(), "pf_halo_send_recv");
    return 0;
}

// every rank receives rank p's `bytes[p]` bytes at recv + offs[p] (uneven parts allowed); `send` = this rank's part
extern "C" int pf_all_gather_v(pf_comm* c, const void* send, void* recv, const long long* bytes, const long long* offs,
                               hipStream_t compute) {
    if (!c || !bytes || !offs) return pf_set_err("pf_all_gather_v: null argument");
    if (order_after(c, compute)) return -1;
    // per part: rank r's part reaches everyone through the windows when it fits a slot (every rank knows every part's size)
    bool via_to[64], via_from[64], rest = false;
    long long sb[64], zo[64];
    for (int p = 0; p < c->world; ++p) {
        sb[p] = bytes[c->rank];
        zo[p] = 0;
        via_to[p] = c->pw.on && (p == c->rank || (bytes[c->rank] > 0 && bytes[c->rank] <= c->pw.slot));
        via_from[p] = c->pw.on && (p == c->rank || (bytes[p] > 0 && bytes[p] <= c->pw.slot));
        rest = rest || (!via_to[p] && bytes[c->rank] > 0) || (!via_from[p] && bytes[p] > 0);
    }
    if (c->pw.on && window_exchange(c, send, sb, zo, recv, bytes, offs, via_to, via_from, "pf_all_gather_v: window transport failed")) return -1;
    if (!rest) return 0;
    if (!c->comm) return pf_set_err("pf_all_gather_v: a part exceeds the window slot and this communicator has no RCCL");
    PF_RCCL(g_api.GroupStart(), "pf_all_gather_v");
    for (int p = 0; p < c->world; ++p) {
        if (bytes[c->rank] > 0 && !via_to[p])
            PF_RCCL_IN_GROUP(g_api.Send(send, (size_t)bytes[c->rank], RCCL_UINT8, p, c->comm, c->stream), "pf_all_gather_v");
        if (bytes[p] > 0 && !via_from[p])
            PF_RCCL_IN_GROUP(g_api.Recv((char*)recv + offs[p], (size_t)bytes[p], RCCL_UINT8, p, c->comm, c->stream), "pf_all_gather_v");
    }
    PF_RCCL(g_api.GroupEnd(), "pf_all_gather_v");
    return 0;
}

extern "C" int pf_all_reduce_sum_f32(pf_comm* c, float* buf, long long count, hipStream_t compute) {
    if (!c || !buf || count < 0) return pf_set_err("pf_all_reduce_sum_f32: bad arguments");
    if (!c->comm) return pf_set_err("pf_all_reduce_sum_f32: this communicator has no RCCL (pf_comm_init_local)");
    if (order_after(c, compute)) return -1;
    PF_RCCL(g_api.AllReduce(buf, buf, (size_t)count, RCCL_FLOAT32, RCCL_SUM, c->comm, c->stream), "pf_all_reduce_sum_f32");
    return 0;
}

extern "C" int pf_broadcast_bytes(pf_comm* c, void* buf, long long bytes, int root, hipStream_t compute) {
    if (!c || !buf || bytes < 0 || root < 0 || root >= c->world) return pf_set_err("pf_broadcast_bytes: bad arguments");
    if (!c->comm) return pf_set_err("pf_broadcast_bytes: this communicator has no RCCL (pf_comm_init_local)");
    if (order_after(c, compute)) return -1;
    PF_RCCL(g_api.Broadcast(buf, buf, (size_t)bytes, RCCL_UINT8, root, c->comm, c->stream), "pf_broadcast_bytes");
    return 0;
}

// `stream` waits (on the device) for everything queued on the communicator so far
extern "C" int pf_comm_wait(pf_comm* c, hipStream_t stream) {
    if (!c) return pf_set_err("pf_comm_wait: null communicator");
    if (hipEventRecord(c->ev_out, c->stream) != hipSuccess) return pf_set_err("pf_comm_wait: hipEventRecord failed");
    if (hipStreamWaitEvent(stream, c->ev_out, 0) != hipSuccess) return pf_set_err("pf_comm_wait: hipStreamWaitEvent failed");
    return 0;
}
