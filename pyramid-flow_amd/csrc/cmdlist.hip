// Launch lists: the kernel sequence of one DiT forward recorded once per (unit, stage) and re-issued from C.
//
// The reference evaluates the transformer 10-20 times per pyramid stage with identical shapes, buffers and weights
// (pyramid_dit_for_video_gen_pipeline.py:611-660: only the timestep, i.e. the CONTENT of the modulation vector and of
// the latents, changes).  The Python host pays ~7 us of interpreter + ctypes time per launch, ~300 launches per forward;
// at 8 sequence-parallel ranks that host time is of the order of a rank's device time.  A launch list stores the
// descriptors of that sequence (the same structs / arguments the eager entry points take) and
//   * pf_cmdlist_run re-issues them through the SAME entry points from one C call (two streams: 0 = the caller's
//     compute stream, 1 = its side stream; cross-stream joins are list entries), communicator calls included;
//   * pf_cmdlist_instantiate captures that replay into a hipGraph (lists without communicator entries), after which
//     pf_cmdlist_run is ONE hipGraphLaunch.
// Nothing here computes: a list is bit-identical to the eager sequence by construction.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "pyflow_hip.h"

int pf_set_err(const char* msg);

namespace {

enum Kind { K_GEMM, K_ATTN, K_LN, K_QK, K_VT, K_RELAYOUT, K_COPY_ROWS, K_A2A, K_COMM_WAIT, K_JOIN };
constexpr int MAX_PARTS = 16;

struct LnArgs { const void* x; void* y; const float* shift; const float* scale; int D, B, rows; long long xb, yb; int ldx, ldy, mb; float eps; };
struct QkArgs { void* qkv; int ld; long long bstride; int q_off, k_off; const float *wq, *wk, *wqt, *wkt, *rope; int B, L, Lt, H; float eps, q_scale; int head_stride; };
struct VtArgs { const void* V; void* Vt; int ldv; long long sV, sVb, sVh; int B, H, L, Lp, head_stride; };
struct RelayoutArgs { void* mat; void* chunks; int rows, B, ld; long long mat_bstride; int n; int col0[MAX_PARTS], cols[MAX_PARTS]; long long off[MAX_PARTS]; int to_chunks; };
struct CopyRowsArgs { const void* src; void* dst; int rows, D, ld_src, ld_dst; long long sb, db; int B; };
struct A2AArgs { pf_comm* c; const void* send; void* recv; long long sb[MAX_PARTS], so[MAX_PARTS], rb[MAX_PARTS], ro[MAX_PARTS]; };
struct CommWaitArgs { pf_comm* c; };
struct JoinArgs { int from, event; };      // stream `slot` waits for everything queued so far on stream `from`

struct Cmd {
    int kind, slot;
    union {
        pf_gemm_desc gemm; pf_attn_desc attn; LnArgs ln; QkArgs qk; VtArgs vt; RelayoutArgs rl; CopyRowsArgs cr; A2AArgs a2a;
        CommWaitArgs cw; JoinArgs join;
    };
    Cmd() { std::memset(this, 0, sizeof(*this)); }
};

}  // namespace

struct pf_cmdlist {
    std::vector<Cmd> cmds;
    std::vector<hipEvent_t> events;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool has_comm = false;
};

namespace {

Cmd& push(pf_cmdlist* l, int kind, int slot) {
    l->cmds.emplace_back();
    Cmd& c = l->cmds.back();
    c.kind = kind;
    c.slot = slot;
    return c;
}

void drop_graph(pf_cmdlist* l) {
    if (l->exec) { hipGraphExecDestroy(l->exec); l->exec = nullptr; }
    if (l->graph) { hipGraphDestroy(l->graph); l->graph = nullptr; }
}

int issue(pf_cmdlist* l, hipStream_t s0, hipStream_t s1) {
    hipStream_t st[2] = {s0, s1};
    for (const Cmd& c : l->cmds) {
        hipStream_t s = st[c.slot];
        int rc = 0;
        switch (c.kind) {
            case K_GEMM: rc = pf_gemm_bf16(&c.gemm, s); break;
            case K_ATTN: rc = pf_attention_bf16(&c.attn, s); break;
            case K_LN:
                rc = pf_ln_modulate(c.ln.x, c.ln.y, c.ln.shift, c.ln.scale, c.ln.D, c.ln.B, c.ln.rows, c.ln.xb, c.ln.yb, c.ln.ldx,
                                    c.ln.ldy, c.ln.mb, c.ln.eps, s);
                break;
            case K_QK:
                rc = pf_qk_norm_rope(c.qk.qkv, c.qk.ld, c.qk.bstride, c.qk.q_off, c.qk.k_off, c.qk.wq, c.qk.wk, c.qk.wqt, c.qk.wkt,
                                     c.qk.rope, c.qk.B, c.qk.L, c.qk.Lt, c.qk.H, c.qk.eps, c.qk.q_scale, c.qk.head_stride, s);
                break;
            case K_VT:
                rc = pf_v_transpose(c.vt.V, c.vt.Vt, c.vt.ldv, c.vt.sV, c.vt.sVb, c.vt.sVh, c.vt.B, c.vt.H, c.vt.L, c.vt.Lp,
                                    c.vt.head_stride, s);
                break;
            case K_RELAYOUT:
                rc = pf_sp_relayout(c.rl.mat, c.rl.chunks, c.rl.rows, c.rl.B, c.rl.ld, c.rl.mat_bstride, c.rl.n, c.rl.col0,
                                    c.rl.cols, c.rl.off, c.rl.to_chunks, s);
                break;
            case K_COPY_ROWS:
                rc = pf_copy_rows(c.cr.src, c.cr.dst, c.cr.rows, c.cr.D, c.cr.ld_src, c.cr.ld_dst, c.cr.sb, c.cr.db, c.cr.B, s);
                break;
            case K_A2A: rc = pf_all_to_all_v(c.a2a.c, c.a2a.send, c.a2a.sb, c.a2a.so, c.a2a.recv, c.a2a.rb, c.a2a.ro, s); break;
            case K_COMM_WAIT: rc = pf_comm_wait(c.cw.c, s); break;
            case K_JOIN: {
                hipEvent_t e = l->events[c.join.event];
                if (hipEventRecord(e, st[c.join.from]) != hipSuccess || hipStreamWaitEvent(s, e, 0) != hipSuccess)
                    return pf_set_err("pf_cmdlist_run: stream join failed");
                break;
            }
            default: return pf_set_err("pf_cmdlist_run: corrupt list");
        }
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

extern "C" {

pf_cmdlist* pf_cmdlist_create(void) { return new pf_cmdlist(); }

int pf_cmdlist_destroy(pf_cmdlist* l) {
    if (!l) return 0;
    drop_graph(l);
    for (hipEvent_t e : l->events) hipEventDestroy(e);
    delete l;
    return 0;
}

int pf_cmdlist_clear(pf_cmdlist* l) {
    if (!l) return pf_set_err("pf_cmdlist_clear: null list");
    drop_graph(l);
    l->cmds.clear();
    l->has_comm = false;
    return 0;
}

int pf_cmdlist_size(const pf_cmdlist* l) { return l ? (int)l->cmds.size() : 0; }
int pf_cmdlist_is_graph(const pf_cmdlist* l) { return l && l->exec ? 1 : 0; }

#define PF_REC_CHECK(l, slot, name)                                            \
    if (!(l)) return pf_set_err(name ": null list");                           \
    if ((slot) < 0 || (slot) > 1) return pf_set_err(name ": slot must be 0 or 1"); \
    if ((l)->exec) return pf_set_err(name ": list is already instantiated")

int pf_cmdlist_gemm(pf_cmdlist* l, const pf_gemm_desc* d, int slot) {
    PF_REC_CHECK(l, slot, "pf_cmdlist_gemm");
    if (!d) return pf_set_err("pf_cmdlist_gemm: null descriptor");
    push(l, K_GEMM, slot).gemm = *d;
    return 0;
}

int pf_cmdlist_attention(pf_cmdlist* l, const pf_attn_desc* d, int slot) {
    PF_REC_CHECK(l, slot, "pf_cmdlist_attention");
    if (!d) return pf_set_err("pf_cmdlist_attention: null descriptor");
    push(l, K_ATTN, slot).attn = *d;
    return 0;
}

int pf_cmdlist_ln_modulate(pf_cmdlist* l, const void* x, void* y, const float* shift, const float* scale, int D, int B,
                           int rows_per_batch, long long x_bstride, long long y_bstride, int ldx, int ldy, int mod_bstride,
                           float eps, int slot) {
    PF_REC_CHECK(l, slot, "pf_cmdlist_ln_modulate");
    push(l, K_LN, slot).ln = LnArgs{x, y, shift, scale, D, B, rows_per_batch, x_bstride, y_bstride, ldx, ldy, mod_bstride, eps};
    return 0;
}

int pf_cmdlist_qk_norm_rope(pf_cmdlist* l, void* qkv, int ld, long long bstride, int q_off, int k_off, const float* wq_img,
                            const float* wk_img, const float* wq_txt, const float* wk_txt, const float* rope, int B, int L,
                            int Lt, int H, float eps, float q_scale, int head_stride, int slot) {
    PF_REC_CHECK(l, slot, "pf_cmdlist_qk_norm_rope");
    push(l, K_QK, slot).qk = QkArgs{qkv, ld, bstride, q_off, k_off, wq_img, wk_img, wq_txt, wk_txt, rope, B, L, Lt, H, eps,
                                    q_scale, head_stride};
    return 0;
}

int pf_cmdlist_v_transpose(pf_cmdlist* l, const void* V, void* Vt, int ldv, long long strideV, long long strideVt_b,
                           long long strideVt_h, int B, int H, int L, int Lp, int head_stride, int slot) {
    PF_REC_CHECK(l, slot, "pf_cmdlist_v_transpose");
    push(l, K_VT, slot).vt = VtArgs{V, Vt, ldv, strideV, strideVt_b, strideVt_h, B, H, L, Lp, head_stride};
    return 0;
}

int pf_cmdlist_sp_relayout(pf_cmdlist* l, void* mat, void* chunks, int rows, int B, int ld, long long mat_bstride, int n_parts,
                           const int* col0, const int* cols, const long long* off, int to_chunks, int slot) {
    PF_REC_CHECK(l, slot, "pf_cmdlist_sp_relayout");
    if (n_parts < 0 || n_parts > MAX_PARTS) return pf_set_err("pf_cmdlist_sp_relayout: at most 16 parts");
    RelayoutArgs& a = push(l, K_RELAYOUT, slot).rl;
    a.mat = mat; a.chunks = chunks; a.rows = rows; a.B = B; a.ld = ld; a.mat_bstride = mat_bstride; a.n = n_parts;
    a.to_chunks = to_chunks;
    for (int i = 0; i < n_parts; ++i) { a.col0[i] = col0[i]; a.cols[i] = cols[i]; a.off[i] = off[i]; }
    return 0;
}

int pf_cmdlist_copy_rows(pf_cmdlist* l, const void* src, void* dst, int rows, int D, int ld_src, int ld_dst,
                         long long src_bstride, long long dst_bstride, int B, int slot) {
    PF_REC_CHECK(l, slot, "pf_cmdlist_copy_rows");
    push(l, K_COPY_ROWS, slot).cr = CopyRowsArgs{src, dst, rows, D, ld_src, ld_dst, src_bstride, dst_bstride, B};
    return 0;
}

int pf_cmdlist_all_to_all_v(pf_cmdlist* l, pf_comm* c, const void* send, const long long* send_bytes,
                            const long long* send_offs, void* recv, const long long* recv_bytes, const long long* recv_offs,
                            int slot) {
    PF_REC_CHECK(l, slot, "pf_cmdlist_all_to_all_v");
    if (!c) return pf_set_err("pf_cmdlist_all_to_all_v: null communicator");
    const int P = pf_comm_world(c);
    if (P > MAX_PARTS) return pf_set_err("pf_cmdlist_all_to_all_v: at most 16 ranks");
    A2AArgs& a = push(l, K_A2A, slot).a2a;
    a.c = c; a.send = send; a.recv = recv;
    for (int i = 0; i < P; ++i) { a.sb[i] = send_bytes[i]; a.so[i] = send_offs[i]; a.rb[i] = recv_bytes[i]; a.ro[i] = recv_offs[i]; }
    l->has_comm = true;
    return 0;
}

int pf_cmdlist_comm_wait(pf_cmdlist* l, pf_comm* c, int slot) {
    PF_REC_CHECK(l, slot, "pf_cmdlist_comm_wait");
    if (!c) return pf_set_err("pf_cmdlist_comm_wait: null communicator");
    push(l, K_COMM_WAIT, slot).cw = CommWaitArgs{c};
    l->has_comm = true;
    return 0;
}

int pf_cmdlist_join(pf_cmdlist* l, int from_slot, int to_slot) {
    PF_REC_CHECK(l, to_slot, "pf_cmdlist_join");
    if (from_slot < 0 || from_slot > 1 || from_slot == to_slot) return pf_set_err("pf_cmdlist_join: bad stream slots");
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return pf_set_err("pf_cmdlist_join: hipEventCreate failed");
    l->events.push_back(e);
    push(l, K_JOIN, to_slot).join = JoinArgs{from_slot, (int)l->events.size() - 1};
    return 0;
}

int pf_cmdlist_run(pf_cmdlist* l, pf_stream_t compute, pf_stream_t side) {
    if (!l) return pf_set_err("pf_cmdlist_run: null list");
    if (l->exec) {
        if (hipGraphLaunch(l->exec, (hipStream_t)compute) != hipSuccess) return pf_set_err("pf_cmdlist_run: hipGraphLaunch failed");
        return 0;
    }
    return issue(l, (hipStream_t)compute, (hipStream_t)side);
}

int pf_cmdlist_instantiate(pf_cmdlist* l, pf_stream_t compute, pf_stream_t side) {
    if (!l) return pf_set_err("pf_cmdlist_instantiate: null list");
    if (l->exec) return 0;
    if (l->has_comm) return pf_set_err("pf_cmdlist_instantiate: lists with communicator entries are replayed, not captured");
    hipStream_t s0 = (hipStream_t)compute, s1 = (hipStream_t)side;
    if (hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal) != hipSuccess)
        return pf_set_err("pf_cmdlist_instantiate: hipStreamBeginCapture failed");
    const int rc = issue(l, s0, s1);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s0, &g);
    if (rc || e != hipSuccess || !g) {
        if (g) hipGraphDestroy(g);
        (void)hipGetLastError();
        return rc ? rc : pf_set_err("pf_cmdlist_instantiate: hipStreamEndCapture failed (a side-stream entry not joined back?)");
    }
    hipGraphExec_t x = nullptr;
    if (hipGraphInstantiate(&x, g, nullptr, nullptr, 0) != hipSuccess) {
        hipGraphDestroy(g);
        (void)hipGetLastError();
        return pf_set_err("pf_cmdlist_instantiate: hipGraphInstantiate failed");
    }
    l->graph = g;
    l->exec = x;
    return 0;
}

}  // extern "C"
