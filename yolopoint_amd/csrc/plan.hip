// Error plumbing + the execution plan (an immutable launch list replayed natively, optionally
// through a captured hipGraph).  The plan is what replaces the reference's Python
// module-by-module dispatch (models/YOLOPoint.py:198-246): the host walks the module tree once,
// emits descriptors, and every later forward is one C call.
#include "yp_internal.h"
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>
#include <cstring>

namespace {
thread_local char g_err[512] = "";
}

void yp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* yp_last_error(void) { return g_err; }
extern "C" int yp_version(void) { return 100; }
extern "C" int yp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

enum OpKind { OP_CONV = 0, OP_SPPF = 1, OP_L2NORM = 2, OP_DETECT = 3, OP_GENERIC = 4, OP_CALLBACK = 5 };

struct PlanOp {
    int kind;
    YpConvDesc conv;
    YpDetectDesc det;
    bool has_det = false;
    YpOpArgs gen;
    YpView v[4];
    int B, dtype, C, na, no, rows_total, row_offset;
    float stride;
    float anchors[16];
    float* x_out;
    float* z_out;
    bool has_deps = false;
    std::vector<int> deps;      // earlier op indices this op must wait for (true data dependencies)
    yp_plan_callback_t cb = nullptr;    // OP_CALLBACK: the caller's function enqueues its own launches on the op's stream
    void* cb_user = nullptr;
    int lane = YP_LANE_MAIN;    // YP_LANE_SIDE: runs beside the ops that follow it; YP_LANE_JOIN: waits for every side op first
};

struct YpPlan {
    std::vector<PlanOp> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool parallel = false;              // graph edges rewired to the data dependencies
    hipStream_t side = nullptr;         // second lane (yp_plan_set_lane): stream + fork / join events, created on first use
    hipEvent_t fork = nullptr, join = nullptr;
};

static int run_op(const PlanOp& op, hipStream_t st) {
    switch (op.kind) {
        case OP_CONV: return yp_conv2d_launch(&op.conv, op.has_det ? &op.det : nullptr, st);
        case OP_SPPF: return yp_sppf_pool(op.v[0], op.v[1], op.v[2], op.v[3], op.B, op.dtype, st);
        case OP_L2NORM: return yp_l2norm_f32(op.v[0], op.v[1], op.B, op.C, st);
        case OP_DETECT:
            return yp_detect_decode(op.v[0], op.B, op.na, op.no, op.stride, op.anchors, op.x_out, op.z_out, op.rows_total,
                                    op.row_offset, st);
    }
    if (op.kind == OP_GENERIC) return yp_run_op(&op.gen, st);
    if (op.kind == OP_CALLBACK) {
        const int rc = op.cb(op.cb_user, (void*)st);
        if (rc != YP_OK) yp_set_error("plan: a callback op returned %d", rc);
        return rc;
    }
    yp_set_error("plan: unknown op kind %d", op.kind);
    return YP_ERR_INVALID;
}

extern "C" int yp_plan_create(YpPlan** plan) {
    YP_REQUIRE(plan != nullptr, "yp_plan_create: null out pointer");
    *plan = new (std::nothrow) YpPlan();
    YP_REQUIRE(*plan != nullptr, "yp_plan_create: out of host memory");
    return YP_OK;
}

extern "C" int yp_plan_destroy(YpPlan* plan) {
    if (!plan) return YP_OK;
    if (plan->exec) (void)hipGraphExecDestroy(plan->exec);
    if (plan->graph) (void)hipGraphDestroy(plan->graph);
    if (plan->fork) (void)hipEventDestroy(plan->fork);
    if (plan->join) (void)hipEventDestroy(plan->join);
    delete plan;
    return YP_OK;
}

#define YP_PLAN_MUTABLE(plan)                                                            \
    YP_REQUIRE((plan) != nullptr, "plan: null handle");                                   \
    YP_REQUIRE((plan)->exec == nullptr, "plan: immutable after yp_plan_instantiate_graph")

extern "C" int yp_plan_add_conv(YpPlan* plan, const YpConvDesc* d) {
    YP_PLAN_MUTABLE(plan);
    YP_REQUIRE(d != nullptr, "yp_plan_add_conv: null descriptor");
    PlanOp op{};
    op.kind = OP_CONV;
    op.conv = *d;
    plan->ops.push_back(op);
    return YP_OK;
}

extern "C" int yp_plan_add_conv_detect(YpPlan* plan, const YpConvDesc* d, const YpDetectDesc* det) {
    YP_PLAN_MUTABLE(plan);
    YP_REQUIRE(d != nullptr && det != nullptr, "yp_plan_add_conv_detect: null descriptor");
    PlanOp op{};
    op.kind = OP_CONV;
    op.conv = *d;
    op.det = *det;
    op.has_det = true;
    plan->ops.push_back(op);
    return YP_OK;
}

extern "C" int yp_plan_add_sppf_pool(YpPlan* plan, YpView x, YpView y1, YpView y2, YpView y3, int B, int dtype) {
    YP_PLAN_MUTABLE(plan);
    PlanOp op{};
    op.kind = OP_SPPF;
    op.v[0] = x; op.v[1] = y1; op.v[2] = y2; op.v[3] = y3;
    op.B = B; op.dtype = dtype;
    plan->ops.push_back(op);
    return YP_OK;
}

extern "C" int yp_plan_add_l2norm(YpPlan* plan, YpView in, YpView out, int B, int C) {
    YP_PLAN_MUTABLE(plan);
    PlanOp op{};
    op.kind = OP_L2NORM;
    op.v[0] = in; op.v[1] = out;
    op.B = B; op.C = C;
    plan->ops.push_back(op);
    return YP_OK;
}

extern "C" int yp_plan_add_detect_decode(YpPlan* plan, YpView raw, int B, int na, int no, float stride,
                                         const float* anchors_px_host, float* x_out, float* z_out, int rows_total,
                                         int row_offset) {
    YP_PLAN_MUTABLE(plan);
    YP_REQUIRE(anchors_px_host && na > 0 && na <= 8, "yp_plan_add_detect_decode: bad anchors");
    PlanOp op{};
    op.kind = OP_DETECT;
    op.v[0] = raw;
    op.B = B; op.na = na; op.no = no; op.stride = stride;
    for (int i = 0; i < na * 2; ++i) op.anchors[i] = anchors_px_host[i];
    op.x_out = x_out; op.z_out = z_out; op.rows_total = rows_total; op.row_offset = row_offset;
    plan->ops.push_back(op);
    return YP_OK;
}

extern "C" int yp_plan_add_op(YpPlan* plan, const YpOpArgs* a) {
    YP_PLAN_MUTABLE(plan);
    YP_REQUIRE(a != nullptr, "yp_plan_add_op: null args");
    PlanOp op{};
    op.kind = OP_GENERIC;
    op.gen = *a;
    plan->ops.push_back(op);
    return YP_OK;
}

// A callback op: when the replay reaches it, `fn(user, stream)` is called on the host and may enqueue launches of its own on `stream` (the
// lane's stream).  For work that belongs INSIDE the schedule of a plan without being a plan op -- the frame pipeline's keypoint
// post-processing, which only needs the keypoint head and runs on the side lane beside the rest of the forward.  Plans with callback ops
// replay eagerly (a capture would freeze the launches the callback made at capture time).
extern "C" int yp_plan_add_callback(YpPlan* plan, yp_plan_callback_t fn, void* user) {
    YP_PLAN_MUTABLE(plan);
    YP_REQUIRE(fn != nullptr, "yp_plan_add_callback: null function");
    PlanOp op{};
    op.kind = OP_CALLBACK;
    op.cb = fn;
    op.cb_user = user;
    plan->ops.push_back(op);
    return YP_OK;
}

extern "C" int yp_plan_num_ops(const YpPlan* plan) { return plan ? (int)plan->ops.size() : 0; }

// Replace view `slot` of generic op `op` (yp_plan_add_op) before the plan is instantiated: a builder that learns only AFTER it has emitted an op
// that one of its outputs has no reader (fp8 training: the 16-bit copy of a BatchNorm output all of whose consumers read the 1-byte twin)
// clears that view's pointer.
extern "C" int yp_plan_patch_op_view(YpPlan* plan, int op, int slot, YpView v) {
    YP_PLAN_MUTABLE(plan);
    YP_REQUIRE(op >= 0 && op < (int)plan->ops.size() && slot >= 0 && slot < 4 && plan->ops[op].kind == OP_GENERIC, "yp_plan_patch_op_view: op %d / view %d is not a view of a generic op", op, slot);
    plan->ops[op].gen.v[slot] = v;
    return YP_OK;
}

// ---------------------------------------------------------------------------------------------
// Streams that really run beside the caller's.  The runtime multiplexes all HIP streams of a process onto a handful of hardware queues
// (four by default); two streams that land on the same queue execute strictly one after the other, and which queue a stream gets depends
// on how many streams the process (PyTorch's pool included) has created before it.  Measured: a second TrainStep in one process put the
// plans' side lane on the main stream's queue -- the two-lane forward then ran SLOWER than one lane (9.5 vs 7.4 ms per step), and with
// GPU_MAX_HW_QUEUES=2 the inference forward fell back to its one-lane time.  So the library keeps a small pool of streams (created once per
// device, never destroyed: see below) and, the first time a caller's stream asks for a companion, TESTS candidates: a 200 us spin kernel on
// the caller's stream, an empty kernel on the candidate -- when the empty kernel's event completes while the spin's has not, the two
// streams are on different queues.  slot 0 = the plans' side lane, slot 1 = an auxiliary stream (engine.TrainStep's loss / label stream);
// the picks of one caller are tested against each other too.
//
// (One side stream per plan, destroyed with the plan, left the runtime in a state in which a later, unrelated hipGraphLaunch crashed --
// ROCm 7.2, reproducible only after ~400 tests in one process.  Sharing is harmless: a plan orders its side ops with its own events.)
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int POOL_STREAMS = 8, PICK_SLOTS = 2;
struct StreamSet { hipStream_t main; hipStream_t pick[PICK_SLOTS]; };
struct DevStreams { std::vector<hipStream_t> pool; std::vector<StreamSet> sets; };
DevStreams g_dev_streams[64];
std::mutex g_side_mutex;

__global__ void yp_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
__global__ void yp_nop_kernel() {}

// true when work on `b` completes while `a` is still busy
bool streams_concurrent(hipStream_t a, hipStream_t b) {
    hipEvent_t ea = nullptr, eb = nullptr;
    if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return true;
    }
    yp_spin_kernel<<<1, 64, 0, a>>>(20000);              // 200 us of the 100 MHz wall clock
    (void)hipEventRecord(ea, a);
    yp_nop_kernel<<<1, 64, 0, b>>>();
    (void)hipEventRecord(eb, b);
    (void)hipEventSynchronize(eb);
    const bool conc = hipEventQuery(ea) == hipErrorNotReady;
    (void)hipGetLastError();
    (void)hipEventSynchronize(ea);
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    return conc;
}
}  // namespace

extern "C" int yp_stream_pick(void* main_stream, int slot, void** out) {
    YP_REQUIRE(out != nullptr && slot >= 0 && slot < PICK_SLOTS, "yp_stream_pick: bad arguments");
    hipStream_t main = (hipStream_t)main_stream;
    int dev = 0;
    YP_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_side_mutex);
    DevStreams& ds = g_dev_streams[dev & 63];
    StreamSet* set = nullptr;
    for (StreamSet& s : ds.sets)
        if (s.main == main) { set = &s; break; }
    if (set == nullptr) {
        ds.sets.push_back(StreamSet{main, {nullptr, nullptr}});
        set = &ds.sets.back();
    }
    if (set->pick[slot] == nullptr) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        const bool capturing = main != nullptr && hipStreamIsCapturing(main, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
        (void)hipGetLastError();
        const bool test = !capturing;
        while ((int)ds.pool.size() < POOL_STREAMS) {
            hipStream_t s = nullptr;
            // (a lowest-priority side stream was measured: no effect on the two-lane forward, 0.687-0.693 vs 0.686-0.702 ms)
            YP_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            ds.pool.push_back(s);
        }
        hipStream_t chosen = nullptr;
        for (hipStream_t cand : ds.pool) {
            bool taken = cand == main;
            for (int k = 0; k < PICK_SLOTS; ++k) taken = taken || set->pick[k] == cand;
            if (taken) continue;
            bool ok = !test || streams_concurrent(main, cand);
            for (int k = 0; k < PICK_SLOTS && ok && test; ++k)
                if (set->pick[k] != nullptr) ok = streams_concurrent(set->pick[k], cand);
            if (getenv("YP_STREAM_DEBUG")) fprintf(stderr, "[yp_stream_pick] main %p slot %d candidate %p: %s\n", (void*)main, slot, (void*)cand, ok ? "concurrent" : "shares a queue");
            if (ok) { chosen = cand; break; }
        }
        if (chosen == nullptr)                            // nothing passed: pool order
            for (hipStream_t cand : ds.pool) {
                bool taken = cand == main;
                for (int k = 0; k < PICK_SLOTS; ++k) taken = taken || set->pick[k] == cand;
                if (!taken) { chosen = cand; break; }
            }
        if (capturing) { *out = chosen; return YP_OK; }   // (not remembered: decided again, with the test, outside the capture)
        set->pick[slot] = chosen;
    }
    *out = set->pick[slot];
    return YP_OK;
}

// Forget the companions of `main_stream` on the current device (main_stream == (void*)-1: of every stream): a caller that DESTROYS a stream it
// passed to yp_stream_pick (or to a plan replay with a side lane) calls this, because a later stream may reuse the handle's address and would
// inherit picks that were tested against another hardware-queue assignment.  The pool streams themselves stay (never destroyed, see above).
extern "C" int yp_stream_forget(void* main_stream) {
    int dev = 0;
    YP_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_side_mutex);
    DevStreams& ds = g_dev_streams[dev & 63];
    if (main_stream == (void*)-1) { ds.sets.clear(); return YP_OK; }
    for (size_t i = 0; i < ds.sets.size(); ++i)
        if (ds.sets[i].main == (hipStream_t)main_stream) { ds.sets.erase(ds.sets.begin() + i); break; }
    return YP_OK;
}

static int ensure_side(YpPlan* plan, hipStream_t st) {
    void* s = nullptr;
    if (int rc = yp_stream_pick((void*)st, 0, &s)) return rc;
    plan->side = (hipStream_t)s;
    if (plan->fork == nullptr) {
        YP_CHECK_HIP(hipEventCreateWithFlags(&plan->fork, hipEventDisableTiming));
        YP_CHECK_HIP(hipEventCreateWithFlags(&plan->join, hipEventDisableTiming));
    }
    return YP_OK;
}

// Ops on YP_LANE_SIDE are issued on the plan's second stream: they wait for everything issued before them (fork event) and
// nothing after them waits for them until an op on YP_LANE_JOIN (or the end of the plan).  Under stream capture the events become
// graph edges, so the side ops form a parallel branch of the hipGraph.  Used by the training backward: the weight-gradient
// kernels (one workgroup per CU, atomics-bound) run beside the dgrad / BatchNorm-backward chain, which never reads their output.
static int run_eager_ops(YpPlan* plan, hipStream_t st, bool& pending) {
    pending = false;                    // side work in flight that the main lane has not joined
    bool main_since_fork = true;        // a side op forks again only when the main lane has moved since the last fork: consecutive side ops are
                                        // ordered by their own stream.  (Not only an economy: captured into a hipGraph, a main-lane node with many
                                        // outgoing cross-stream edges -- every side op re-forking from the same position -- lost its SAME-stream
                                        // successor edge at replay on ROCm 7.2: the next main op ran before it.)
    // ENQUEUE order.  A run of side ops waits for the fork event recorded where the run starts in the op list, but its launches are handed to
    // the runtime ALTERNATELY with the main-lane ops that follow (one side op behind each main op; the rest when the next run / join / end
    // comes).  When the host is a whole step ahead of the device the order is irrelevant; when every frame ends in a host read-back (the
    // frame pipeline) the device executes launches almost as they arrive, and a block of ten side launches enqueued in one go left the main
    // lane empty meanwhile: the lanes took turns instead of overlapping.
    std::vector<const PlanOp*> queued;
    auto issue_one = [&]() -> int {
        const PlanOp* q = queued.front();
        queued.erase(queued.begin());
        return run_op(*q, plan->side);
    };
    auto flush = [&]() -> int {
        while (!queued.empty())
            if (int rc = issue_one()) return rc;
        return YP_OK;
    };
    bool have_side = false;
    for (const PlanOp& op : plan->ops) {
        if (op.lane == YP_LANE_SIDE) {
            if (!have_side) {
                if (int rc = ensure_side(plan, st)) return rc;
                have_side = true;
            }
            if (main_since_fork) {
                if (int rc = flush()) return rc;           // (earlier side work keeps its place in the side stream's order)
                YP_CHECK_HIP(hipEventRecord(plan->fork, st));
                YP_CHECK_HIP(hipStreamWaitEvent(plan->side, plan->fork, 0));
                main_since_fork = false;
                if (int rc = run_op(op, plan->side)) return rc;    // the first op of a run goes out at once
            } else {
                queued.push_back(&op);
            }
            pending = true;
            continue;
        }
        main_since_fork = true;
        if (op.lane == YP_LANE_JOIN && pending) {
            if (int rc = flush()) return rc;
            YP_CHECK_HIP(hipEventRecord(plan->join, plan->side));
            YP_CHECK_HIP(hipStreamWaitEvent(st, plan->join, 0));
            pending = false;
        }
        const int rc = run_op(op, st);
        if (rc != YP_OK) return rc;
        if (!queued.empty())
            if (int rc2 = issue_one()) return rc2;
    }
    if (int rc = flush()) return rc;
    if (pending) {
        YP_CHECK_HIP(hipEventRecord(plan->join, plan->side));
        YP_CHECK_HIP(hipStreamWaitEvent(st, plan->join, 0));
    }
    return YP_OK;
}

// On an error return (a callback op that failed, a launch that was rejected) side-lane work already enqueued may still be running: the
// caller's stream is made to wait for it before the error propagates -- the buffers such kernels write (the front end's temporaries, which
// return to the main stream's allocator pool when its hook raises) must not be handed out again under them.  Queued side ops that were not
// issued yet are dropped with the error.
static int run_eager(YpPlan* plan, hipStream_t st) {
    bool pending = false;
    const int rc = run_eager_ops(plan, st, pending);
    if (rc != YP_OK && pending && plan->side != nullptr && plan->join != nullptr) {
        if (hipEventRecord(plan->join, plan->side) == hipSuccess) (void)hipStreamWaitEvent(st, plan->join, 0);
        else (void)hipStreamSynchronize(plan->side);
    }
    return rc;
}

extern "C" int yp_plan_set_lane(YpPlan* plan, int op, int lane) {
    YP_PLAN_MUTABLE(plan);
    YP_REQUIRE(op >= 0 && op < (int)plan->ops.size() && lane >= YP_LANE_MAIN && lane <= YP_LANE_JOIN, "yp_plan_set_lane: bad argument");
    plan->ops[op].lane = lane;
    return YP_OK;
}

extern "C" int yp_plan_set_deps(YpPlan* plan, int op, const int* deps, int ndeps) {
    YP_PLAN_MUTABLE(plan);
    YP_REQUIRE(op >= 0 && op < (int)plan->ops.size() && ndeps >= 0, "yp_plan_set_deps: bad argument");
    PlanOp& o = plan->ops[op];
    o.deps.clear();
    o.has_deps = true;
    for (int i = 0; i < ndeps; ++i) {
        YP_REQUIRE(deps[i] >= 0 && deps[i] < op, "yp_plan_set_deps: dependency %d of op %d is not an earlier op", deps[i], op);
        o.deps.push_back(deps[i]);
    }
    return YP_OK;
}

// Capture the launch list into a hipGraph.  The ops are captured as a linear chain on `stream`
// (one kernel node per op); when every op carries a dependency list the chain's edges are then
// replaced by the true data dependencies, so independent branches of the network (keypoint head,
// descriptor head, YOLO encoder, C3.cv2 beside the bottleneck chain, the three Detect levels) become
// parallel branches of the graph and overlap on the GPU.
extern "C" int yp_plan_instantiate_graph(YpPlan* plan, void* stream) {
    YP_REQUIRE(plan != nullptr, "yp_plan_instantiate_graph: null plan");
    YP_REQUIRE(plan->exec == nullptr, "yp_plan_instantiate_graph: already instantiated");
    hipStream_t st = (hipStream_t)stream;
    YP_REQUIRE(st != nullptr, "yp_plan_instantiate_graph: capture needs a non-default stream");
    for (const PlanOp& op : plan->ops) YP_REQUIRE(op.kind != OP_CALLBACK, "yp_plan_instantiate_graph: a plan with callback ops replays eagerly");
    const size_t n = plan->ops.size();
    for (const PlanOp& op : plan->ops)
        if (op.lane == YP_LANE_SIDE) { if (int rc = ensure_side(plan, st)) return rc; break; }     // (no stream creation inside a capture)
    bool rewire = n > 1;
    for (const PlanOp& op : plan->ops) rewire = rewire && op.has_deps && op.lane == YP_LANE_MAIN;
    YP_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = YP_OK;
    // last[i] = the last graph node op i produced (an op may launch several kernels, or none): read off the capture's frontier after
    // every op.  Plans with schedule lanes (side streams) keep the captured topology.
    std::vector<hipGraphNode_t> last(n, nullptr);
    if (rewire) {
        for (size_t i = 0; i < n && rc == YP_OK; ++i) {
            rc = run_op(plan->ops[i], st);
            if (rc != YP_OK) break;
            hipStreamCaptureStatus status;
            const hipGraphNode_t* frontier = nullptr;
            size_t nf = 0;
            if (hipStreamGetCaptureInfo_v2(st, &status, nullptr, nullptr, &frontier, &nf) != hipSuccess || status != hipStreamCaptureStatusActive || nf > 1) {
                rewire = false;                     // (unexpected capture state: fall back to the linear chain)
                for (size_t k = i + 1; k < n && rc == YP_OK; ++k) rc = run_op(plan->ops[k], st);
                break;
            }
            last[i] = nf == 1 ? frontier[0] : (i ? last[i - 1] : nullptr);
        }
    } else {
        rc = run_eager(plan, st);
    }
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != YP_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    YP_CHECK_HIP(e);
    plan->graph = g;

    size_t nn = 0, ne = 0;
    YP_CHECK_HIP(hipGraphGetNodes(g, nullptr, &nn));
    YP_CHECK_HIP(hipGraphGetEdges(g, nullptr, nullptr, &ne));
    if (rewire && nn >= 1 && ne == nn - 1) {
        // the capture is a linear chain of nn nodes; op i owns the run of nodes behind last[i-1] up to last[i] (possibly empty)
        std::vector<hipGraphNode_t> from(ne), to(ne);
        if (ne) YP_CHECK_HIP(hipGraphGetEdges(g, from.data(), to.data(), &ne));
        size_t nroot = 0;
        YP_CHECK_HIP(hipGraphGetRootNodes(g, nullptr, &nroot));
        std::vector<hipGraphNode_t> chain;
        if (nroot == 1) {
            hipGraphNode_t cur;
            YP_CHECK_HIP(hipGraphGetRootNodes(g, &cur, &nroot));
            chain.push_back(cur);
            for (size_t step = 0; step + 1 < nn; ++step) {
                bool found = false;
                for (size_t k = 0; k < ne && !found; ++k)
                    if (from[k] == cur) { cur = to[k]; found = true; }
                if (!found) break;
                chain.push_back(cur);
            }
        }
        if (chain.size() == nn) {
            std::vector<long> first_pos(n, -1), last_pos(n, -1);      // positions in `chain` of op i's first / last node (-1: the op has no node)
            long pos = 0;
            bool ok = true;
            for (size_t i = 0; i < n && ok; ++i) {
                const hipGraphNode_t prev_last = i ? last[i - 1] : nullptr;
                if (last[i] == prev_last) continue;                   // no node of its own
                long end = pos;
                while (end < (long)nn && chain[end] != last[i]) ++end;
                if (end >= (long)nn) { ok = false; break; }
                first_pos[i] = pos; last_pos[i] = end;
                pos = end + 1;
            }
            if (ok && pos == (long)nn) {
                // drop the edges BETWEEN ops (keep the chains inside multi-kernel ops), then add the data dependencies.  A dependency on
                // an op without nodes is inherited from that op's own dependencies.
                std::vector<hipGraphNode_t> rf, rt;
                for (size_t i = 0; i < n; ++i)
                    if (first_pos[i] > 0) { rf.push_back(chain[first_pos[i] - 1]); rt.push_back(chain[first_pos[i]]); }
                if (!rf.empty()) YP_CHECK_HIP(hipGraphRemoveDependencies(g, rf.data(), rt.data(), rf.size()));
                std::vector<std::vector<int>> eff(n);
                for (size_t j = 0; j < n; ++j) {
                    std::vector<int> stack(plan->ops[j].deps.begin(), plan->ops[j].deps.end());
                    std::vector<char> seen(n, 0);
                    while (!stack.empty()) {
                        const int d = stack.back(); stack.pop_back();
                        if (seen[d]) continue;
                        seen[d] = 1;
                        if (first_pos[d] >= 0) eff[j].push_back(d);
                        else for (int dd : plan->ops[d].deps) stack.push_back(dd);
                    }
                }
                for (size_t j = 0; j < n; ++j) {
                    if (first_pos[j] < 0) continue;
                    for (int d : eff[j]) YP_CHECK_HIP(hipGraphAddDependencies(g, &chain[last_pos[d]], &chain[first_pos[j]], 1));
                }
                plan->parallel = true;
            }
        }
    }
    if (getenv("YP_GRAPH_DUMP")) {          // debug: the captured topology as node index -> indices of the nodes it waits for
        std::vector<hipGraphNode_t> nodes(nn);
        size_t cnt = nn;
        if (nn && hipGraphGetNodes(g, nodes.data(), &cnt) == hipSuccess) {
            for (size_t i = 0; i < cnt; ++i) {
                size_t nd = 0;
                (void)hipGraphNodeGetDependencies(nodes[i], nullptr, &nd);
                std::vector<hipGraphNode_t> deps(nd ? nd : 1);
                if (nd) (void)hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd);
                hipGraphNodeType ty;
                (void)hipGraphNodeGetType(nodes[i], &ty);
                fprintf(stderr, "[graph] node %zu type %d <-", i, (int)ty);
                for (size_t k = 0; k < nd; ++k)
                    for (size_t j = 0; j < cnt; ++j)
                        if (nodes[j] == deps[k]) fprintf(stderr, " %zu", j);
                fprintf(stderr, "\n");
            }
        }
    }
    for (const PlanOp& op : plan->ops)
        if (op.lane == YP_LANE_SIDE) plan->parallel = true;     // (schedule lanes: the captured topology has a side branch)
    YP_CHECK_HIP(hipGraphInstantiate(&plan->exec, g, nullptr, nullptr, 0));
    return YP_OK;
}

extern "C" int yp_plan_graph_is_parallel(const YpPlan* plan) { return plan && plan->parallel ? 1 : 0; }

extern "C" int yp_plan_run(YpPlan* plan, void* stream) {
    YP_REQUIRE(plan != nullptr, "yp_plan_run: null plan");
    hipStream_t st = (hipStream_t)stream;
    if (plan->exec) {
        YP_CHECK_HIP(hipGraphLaunch(plan->exec, st));
        return YP_OK;
    }
    return run_eager(plan, st);
}

extern "C" int yp_plan_profile(YpPlan* plan, void* stream, float* ms_out_host) {
    YP_REQUIRE(plan != nullptr && ms_out_host != nullptr, "yp_plan_profile: null argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = plan->ops.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) YP_CHECK_HIP(hipEventCreate(&e));
    int rc = YP_OK;
    YP_CHECK_HIP(hipEventRecord(ev[0], st));
    for (size_t i = 0; i < n && rc == YP_OK; ++i) {
        rc = run_op(plan->ops[i], st);
        if (rc == YP_OK && hipEventRecord(ev[i + 1], st) != hipSuccess) rc = YP_ERR_HIP;
    }
    if (rc == YP_OK && hipStreamSynchronize(st) != hipSuccess) rc = YP_ERR_HIP;
    if (rc == YP_OK)
        for (size_t i = 0; i < n; ++i)
            if (hipEventElapsedTime(&ms_out_host[i], ev[i], ev[i + 1]) != hipSuccess) rc = YP_ERR_HIP;
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (rc == YP_ERR_HIP) yp_set_error("yp_plan_profile: HIP event failure");
    return rc;
}

extern "C" int yp_plan_time(YpPlan* plan, void* stream, int iters, float* ms_per_iter_host) {
    YP_REQUIRE(plan != nullptr && ms_per_iter_host != nullptr && iters > 0, "yp_plan_time: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    YP_CHECK_HIP(hipEventCreate(&e0));
    YP_CHECK_HIP(hipEventCreate(&e1));
    YP_CHECK_HIP(hipEventRecord(e0, st));
    int rc = YP_OK;
    for (int i = 0; i < iters && rc == YP_OK; ++i) rc = yp_plan_run(plan, st);
    if (rc == YP_OK) {
        YP_CHECK_HIP(hipEventRecord(e1, st));
        YP_CHECK_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        YP_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        *ms_per_iter_host = ms / (float)iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
