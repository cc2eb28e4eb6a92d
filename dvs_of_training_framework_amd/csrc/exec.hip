// Step executor: a captured training (or inference) step replayed as PLAIN
// kernel launches from one C call.
//
// The loop body of the reference (utils/training.py:138-167) is ~120 kernel
// launches here.  Enqueued from Python it costs the host ~2 ms per step
// (ctypes + descriptor building, ~17 us per launch); replayed by
// hipGraphLaunch the host is free, but ROCm 7.2's graph runtime re-schedules
// the two backward branches onto its own queues (the hand-tuned two-stream
// overlap is lost: 3.47 vs 3.24 ms of GPU time per step).  This
// executor keeps the capture -- one recording of the step for a batch
// signature -- and drops the graph runtime: it reads the kernel nodes and the
// edges of the captured hipGraph, splits the nodes into "lanes" (one per HIP
// stream), and replays them in capture order with hipLaunchKernel, an event
// per edge that crosses lanes.  Same kernels, same arguments, same order as
// the eager step: bit-identical results, the two-stream overlap, and a host
// cost of a bare launch call per kernel.
//
// Lanes.  A dependency that crosses queues costs the consumer ~13 us after the
// producer has ended (measured; same queue: 0-7 us), so the chain that bounds
// the step must never change queues.  dvsof_exec_calibrate runs the step once
// on ONE stream with a timing event between kernels; lane 0 (the caller's
// stream) is then the longest path of the DAG by measured time, lane 1 the
// longest path among the remaining nodes, ..., the last lane takes whatever
// is left, in capture order.  Before calibration a greedy chain split is used
// (extend the chain of a dependency that is still the tail of its lane).
//
// The graph must consist of kernel nodes (and empty nodes); it has to stay
// alive while the executor is (kernel arguments are read from the nodes).
//
// Data parallelism.  The gradient exchange is not a kernel of ours: while a
// step is captured parallel.GradReducer leaves MARKS in the capture instead
// of collectives (dvsof_exec_mark: a one-thread no-op kernel whose arguments
// name the bucket).  The executor recognises the marks by their function
// address and never launches them: a BUCKET mark becomes "record an event on
// the lane that closed the bucket, make the exchange stream wait for it,
// ncclAllReduce (average, in place) on the exchange stream"; the JOIN mark in
// front of the optimizer becomes "the lane waits for the exchange stream".
// Same collectives in the same order on every rank (the launch order is a
// property of the captured graph, identical across replicas).  Without a
// communicator (dvsof_exec_set_comm not called: one GPU) marks are skipped.
#include "common.h"
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <unordered_map>
#include <vector>

namespace {

// kind: DVSOF_MARK_BUCKET / DVSOF_MARK_JOIN.  Never runs under the executor;
// under hipGraphLaunch it is a no-op (and the exchange is then missing: the
// captured step refuses that combination).
__global__ void exec_mark_kernel(int kind, int index, float *bucket, size_t n) {}

struct XNode {
    hipKernelNodeParams kp;
    bool kernel;
    int mark = 0;            // DVSOF_MARK_*: not launched
    int mark_index = 0;
    float *mark_ptr = nullptr;
    size_t mark_n = 0;
    hipEvent_t mark_ev = nullptr;   // BUCKET: the lane's progress the exchange stream waits for
    hipEvent_t xdone_ev = nullptr;  // BUCKET: the exchange stream's progress behind this collective
    bool xdone_used = false;        // a WAIT mark refers to this BUCKET mark (else xdone_ev is not recorded)
    int wait_for = -1;              // WAIT: position of the BUCKET mark whose collective the lane waits for
    bool xlane = false;             // runs on the EXCHANGE stream when there is one (see dvsof_exec_create):
                                    // a kernel captured behind nothing but WAIT marks / such kernels -- a
                                    // bucket's optimizer update -- and those WAIT marks themselves
    int lane_plan = 0;              // the plan's lane (xlane nodes: used when there is no exchange stream)
    int lane;
    int id = 0;              // position in capture (topological) order: what plans are written in
    int lane0 = 0;           // lane of the greedy chain split made at creation
    float us = 1.f;          // measured duration (dvsof_exec_calibrate)
    std::vector<int> deps_id;    // ids of the nodes this one depends on
    std::vector<int> window;     // BUCKET mark: ids of the kernels captured behind it that are NOT
                                 // behind its WAIT / the JOIN mark (they must not touch the bucket)
    std::vector<int> deps;   // their positions in the current launch order
    std::vector<int> wait;   // nodes of other lanes to wait for before the launch
    hipEvent_t ev;           // recorded after the launch when another lane waits for it
};

// A lane plan: launch order (node ids) and lane by node id.
struct Plan {
    const char *name;
    std::vector<int> order, lane;
    float best_us = 3.0e38f;     // fastest measured step under this plan
};

struct Exec {
    std::vector<XNode> nodes;    // in launch order
    std::vector<Plan> cands;     // plans being tried on the steps after calibration
    int trial_next = -1;         // next trial (round-robin over cands), -1: settled
    int trial_cur = -1;          // plan of the step in flight whose time is still to be read
    hipEvent_t t0 = nullptr, t1 = nullptr;
    const char *plan_name = "chain";
    std::vector<hipStream_t> side;   // lanes 1.. (lane 0 is the stream of the launch call)
    std::vector<int> tail;           // last node of every lane
    hipEvent_t fork = nullptr;
    std::vector<hipEvent_t> join;    // per side lane
    int n_kernels = 0, n_events = 0, n_waits = 0, n_marks = 0;
    int max_lanes = 1;
    void *comm = nullptr;            // dvsof_comm_create handle (not owned)
    hipStream_t xstream = nullptr;   // exchange stream (not owned)
    hipEvent_t xdone = nullptr;      // exchange stream's progress at a JOIN mark
    hipStream_t ustream = nullptr;   // UPDATE stream: lane of the xlane nodes (own stream: on the exchange
                                     // stream the updates would sit between the collectives -- under the
                                     // loopback exchange that stream is the step's longest chain)
};

// From lanes to waits and events: a node waits for its dependencies in other
// lanes, minus those an earlier wait of its lane on the same producer lane
// already covers (lanes run in order: waiting for node d covers d's
// predecessors in its lane).
int wire(Exec *x)
{
    const int nn = (int)x->nodes.size();
    // the exchange stream is one more lane (index max_lanes) for the nodes that belong on it
    const bool xl = x->comm && x->xstream && x->ustream;
    for (auto &n : x->nodes) n.lane = (n.xlane && xl) ? x->max_lanes : n.lane_plan;
    int L = 0;
    for (auto &n : x->nodes) L = std::max(L, n.lane + 1);
    x->tail.assign(L, -1);
    for (int i = 0; i < nn; ++i) x->tail[x->nodes[i].lane] = i;
    for (auto &n : x->nodes) {
        if (n.ev) (void)hipEventDestroy(n.ev);
        n.ev = nullptr;
        n.wait.clear();
        n.xdone_used = false;
        n.wait_for = -1;
    }
    for (int i = 0; i < nn; ++i) {      // WAIT marks find their BUCKET mark (same mark index)
        XNode &n = x->nodes[i];
        if (n.mark != DVSOF_MARK_WAIT) continue;
        for (int j = 0; j < nn; ++j)
            if (x->nodes[j].mark == DVSOF_MARK_BUCKET && x->nodes[j].mark_index == n.mark_index) {
                n.wait_for = j;
                x->nodes[j].xdone_used = true;
            }
    }
    x->n_events = x->n_waits = 0;
    std::vector<std::vector<int>> seen(L, std::vector<int>(L, -1));
    for (int i = 0; i < nn; ++i) {
        XNode &n = x->nodes[i];
        std::vector<int> w;
        for (int d : n.deps)
            if (x->nodes[d].lane != n.lane) w.push_back(d);
        std::sort(w.begin(), w.end(), [](int a, int b) { return a > b; });
        for (int d : w) {
            const int pl = x->nodes[d].lane;
            if (seen[n.lane][pl] >= d) continue;
            seen[n.lane][pl] = d;
            n.wait.push_back(d);
        }
        x->n_waits += (int)n.wait.size();
    }
    for (auto &n : x->nodes)
        for (int d : n.wait)
            if (!x->nodes[d].ev) {
                DVSOF_HIP_TRY(hipEventCreateWithFlags(&x->nodes[d].ev, hipEventDisableTiming));
                ++x->n_events;
            }
    if (L > 1 && !x->fork) DVSOF_HIP_TRY(hipEventCreateWithFlags(&x->fork, hipEventDisableTiming));
    while ((int)x->join.size() < L - 1) {
        hipEvent_t e;
        DVSOF_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        x->join.push_back(e);
    }
    return DVSOF_OK;
}

// Plan 1, lanes by measured time: lane l = the longest path (sum of node
// durations) through the nodes no earlier lane took; the last lane takes the
// rest.  Launch order = capture order.
Plan plan_by_time(const Exec *x)
{
    const int nn = (int)x->nodes.size();
    std::vector<char> taken(nn, 0);
    std::vector<int> lane(nn, 0);
    int left = nn;
    for (int i = 0; i < nn; ++i)    // the update lane's nodes take no compute lane (wire() places them)
        if (x->nodes[i].xlane) {
            taken[i] = 1;
            lane[i] = x->max_lanes - 1;
            --left;
        }
    for (int l = 0; l < x->max_lanes && left > 0; ++l) {
        if (l == x->max_lanes - 1) {
            for (int i = 0; i < nn; ++i)
                if (!taken[i]) lane[i] = l;
            break;
        }
        std::vector<double> best(nn, 0.0);
        std::vector<int> from(nn, -1);
        int end = -1;
        for (int i = 0; i < nn; ++i) {
            if (taken[i]) continue;
            double b = 0.0;
            for (int d : x->nodes[i].deps)
                if (!taken[d] && best[d] > b) {
                    b = best[d];
                    from[i] = d;
                }
            best[i] = b + (double)x->nodes[i].us;
            if (end < 0 || best[i] > best[end]) end = i;
        }
        for (int i = end; i >= 0; i = from[i]) {
            lane[i] = l;
            taken[i] = 1;
            --left;
        }
    }
    Plan p;
    p.name = "paths";
    p.order.resize(nn);
    p.lane.resize(nn);
    std::vector<int> by_id(nn);
    for (int i = 0; i < nn; ++i) by_id[x->nodes[i].id] = i;
    for (int id = 0; id < nn; ++id) {
        p.order[id] = id;
        p.lane[id] = lane[by_id[id]];
    }
    return p;
}

// Plan 0: the greedy chain split of dvsof_exec_create, capture order.
Plan plan_chain(const Exec *x)
{
    const int nn = (int)x->nodes.size();
    Plan p;
    p.name = "chain";
    p.order.resize(nn);
    p.lane.resize(nn);
    for (const auto &n : x->nodes) {
        p.order[n.id] = n.id;
        p.lane[n.id] = n.lane0;
    }
    return p;
}

// Plan 2, list scheduling on the measured durations: lanes are in-order
// queues; of the nodes whose dependencies are placed the one with the longest
// remaining path to the end of the step goes next, onto the lane where it can
// start first (a dependency in another lane costs `hop` us on top of its end).
// The chain that bounds the step keeps flowing on one lane; whatever hangs off
// it (weight gradients, folds, prepared forms) fills the other.  Also fixes
// the LAUNCH order: the order nodes were placed in.
Plan plan_by_list(const Exec *x, double hop)
{
    const int nn = (int)x->nodes.size(), L = x->max_lanes;
    std::vector<std::vector<int>> succ(nn);
    std::vector<int> indeg(nn, 0);
    for (int i = 0; i < nn; ++i) {
        indeg[i] = (int)x->nodes[i].deps.size();
        for (int d : x->nodes[i].deps) succ[d].push_back(i);
    }
    {   // marks keep their capture order: every rank issues the same collectives in the
        // same order whatever its own measured durations say
        std::vector<int> marks;
        for (int i = 0; i < nn; ++i)
            if (x->nodes[i].mark) marks.push_back(i);
        std::sort(marks.begin(), marks.end(), [&](int a, int b) { return x->nodes[a].id < x->nodes[b].id; });
        for (size_t k = 1; k < marks.size(); ++k) {
            succ[marks[k - 1]].push_back(marks[k]);
            ++indeg[marks[k]];
        }
    }
    std::vector<double> bl(nn, 0.0);
    for (int i = nn - 1; i >= 0; --i) {     // positions are a topological order
        double b = 0.0;
        for (int s_ : succ[i]) b = std::max(b, bl[s_]);
        bl[i] = b + (double)x->nodes[i].us;
    }
    std::vector<double> fin(nn, 0.0), free_at(L, 0.0);
    std::vector<int> lane(nn, 0), ready;
    for (int i = 0; i < nn; ++i)
        if (!indeg[i]) ready.push_back(i);
    Plan p;
    p.name = "list";
    p.order.reserve(nn);
    p.lane.assign(nn, 0);
    while (!ready.empty()) {
        size_t pick = 0;
        for (size_t k = 1; k < ready.size(); ++k) {
            const int a = ready[k], b = ready[pick];
            if (bl[a] > bl[b] || (bl[a] == bl[b] && a < b)) pick = k;
        }
        const int i = ready[pick];
        ready.erase(ready.begin() + (long)pick);
        if (x->nodes[i].xlane) {    // on the update lane: occupies no compute lane
            double t = 0.0;
            for (int d : x->nodes[i].deps) t = std::max(t, fin[d]);
            lane[i] = L - 1;
            fin[i] = t + (double)x->nodes[i].us;
            p.order.push_back(x->nodes[i].id);
            p.lane[x->nodes[i].id] = L - 1;
            for (int s_ : succ[i])
                if (--indeg[s_] == 0) ready.push_back(s_);
            continue;
        }
        int best_l = 0;
        double best_t = 0.0;
        for (int l = 0; l < L; ++l) {
            double t = free_at[l];
            for (int d : x->nodes[i].deps) t = std::max(t, fin[d] + (lane[d] != l ? hop : 0.0));
            if (l == 0 || t < best_t) {
                best_t = t;
                best_l = l;
            }
        }
        lane[i] = best_l;
        fin[i] = best_t + (double)x->nodes[i].us;
        free_at[best_l] = fin[i];
        p.order.push_back(x->nodes[i].id);
        p.lane[x->nodes[i].id] = best_l;
        for (int s_ : succ[i])
            if (--indeg[s_] == 0) ready.push_back(s_);
    }
    return p;
}

// Put the nodes in the plan's launch order and lanes; dependencies as positions again.
void apply(Exec *x, const Plan &p)
{
    const int nn = (int)x->nodes.size();
    std::vector<int> by_id(nn), newpos(nn);
    for (int i = 0; i < nn; ++i) by_id[x->nodes[i].id] = i;
    std::vector<XNode> nv;
    nv.reserve(nn);
    for (int k = 0; k < nn; ++k) {
        newpos[p.order[k]] = k;
        nv.push_back(std::move(x->nodes[by_id[p.order[k]]]));
    }
    for (auto &n : nv) {
        n.lane = n.lane_plan = p.lane[n.id];
        n.deps.clear();
        for (int d : n.deps_id) n.deps.push_back(newpos[d]);
        std::sort(n.deps.begin(), n.deps.end());
    }
    x->nodes.swap(nv);
    x->plan_name = p.name;
}

void destroy(Exec *x)
{
    for (auto &n : x->nodes) {
        if (n.ev) (void)hipEventDestroy(n.ev);
        if (n.mark_ev) (void)hipEventDestroy(n.mark_ev);
        if (n.xdone_ev) (void)hipEventDestroy(n.xdone_ev);
    }
    if (x->xdone) (void)hipEventDestroy(x->xdone);
    if (x->t0) (void)hipEventDestroy(x->t0);
    if (x->t1) (void)hipEventDestroy(x->t1);
    if (x->fork) (void)hipEventDestroy(x->fork);
    for (auto e : x->join)
        if (e) (void)hipEventDestroy(e);
    delete x;
}

int launch_node(const XNode &n, hipStream_t st)
{
    if (n.kp.kernelParams) {
        DVSOF_HIP_TRY(hipLaunchKernel(n.kp.func, n.kp.gridDim, n.kp.blockDim, n.kp.kernelParams,
                                      n.kp.sharedMemBytes, st));
    } else {
        DVSOF_HIP_TRY(hipModuleLaunchKernel((hipFunction_t)n.kp.func, n.kp.gridDim.x, n.kp.gridDim.y,
                                            n.kp.gridDim.z, n.kp.blockDim.x, n.kp.blockDim.y,
                                            n.kp.blockDim.z, n.kp.sharedMemBytes, st, nullptr,
                                            n.kp.extra));
    }
    return DVSOF_OK;
}

// A mark at its place in the launch order, on the stream `st` of its lane.
int run_mark(Exec *x, XNode &n, hipStream_t st)
{
    if (!x->comm) return DVSOF_OK;   // one GPU: nothing to exchange
    hipStream_t xs = x->xstream ? x->xstream : st;
    if (n.mark == DVSOF_MARK_BUCKET) {
        if (xs != st) {
            DVSOF_HIP_TRY(hipEventRecord(n.mark_ev, st));
            DVSOF_HIP_TRY(hipStreamWaitEvent(xs, n.mark_ev, 0));
        }
        const int rc = dvsof_allreduce_bucket(x->comm, n.mark_ptr, n.mark_n, (void *)xs);
        if (rc) return rc;
        if (xs != st && n.xdone_used) DVSOF_HIP_TRY(hipEventRecord(n.xdone_ev, xs));
        return DVSOF_OK;
    }
    if (n.mark == DVSOF_MARK_WAIT) {
        if (xs != st && n.wait_for >= 0) DVSOF_HIP_TRY(hipStreamWaitEvent(st, x->nodes[n.wait_for].xdone_ev, 0));
        return DVSOF_OK;
    }
    if (n.mark == DVSOF_MARK_JOIN && xs != st) {
        DVSOF_HIP_TRY(hipEventRecord(x->xdone, xs));
        DVSOF_HIP_TRY(hipStreamWaitEvent(st, x->xdone, 0));
    }
    return DVSOF_OK;
}

}  // namespace

extern "C" {

int dvsof_exec_mark(int kind, int index, float *bucket, size_t n, void *stream)
{
    if (kind != DVSOF_MARK_BUCKET && kind != DVSOF_MARK_JOIN && kind != DVSOF_MARK_WAIT) return DVSOF_EINVAL;
    if (kind == DVSOF_MARK_BUCKET && (!bucket || n == 0)) return DVSOF_EINVAL;
    hipLaunchKernelGGL(exec_mark_kernel, dim3(1), dim3(1), 0, as_stream(stream), kind, index, bucket, n);
    DVSOF_LAUNCH_CHECK();
    return DVSOF_OK;
}

int dvsof_exec_set_comm(void *exec, void *comm, void *exchange_stream)
{
    if (!exec) return DVSOF_EINVAL;
    Exec *x = (Exec *)exec;
    x->comm = comm;
    x->xstream = as_stream(exchange_stream);
    if (comm && !x->xdone) DVSOF_HIP_TRY(hipEventCreateWithFlags(&x->xdone, hipEventDisableTiming));
    // (without dvsof_exec_set_update_stream the updates go on the exchange stream itself, between
    // the collectives)
    if (comm && x->xstream && !x->ustream) x->ustream = x->xstream;
    for (auto &n : x->nodes)
        if (comm && n.mark == DVSOF_MARK_BUCKET && !n.mark_ev) {
            DVSOF_HIP_TRY(hipEventCreateWithFlags(&n.mark_ev, hipEventDisableTiming));
            DVSOF_HIP_TRY(hipEventCreateWithFlags(&n.xdone_ev, hipEventDisableTiming));
        }
    return wire(x);     // the exchange stream's lane exists from here on
}

int dvsof_exec_set_update_stream(void *exec, void *update_stream)
{
    if (!exec || !update_stream) return DVSOF_EINVAL;
    Exec *x = (Exec *)exec;
    x->ustream = as_stream(update_stream);
    return wire(x);
}

int dvsof_exec_marks(void *exec, int *n_marks)
{
    if (!exec || !n_marks) return DVSOF_EINVAL;
    *n_marks = ((Exec *)exec)->n_marks;
    return DVSOF_OK;
}

int dvsof_exec_mark_window(void *exec, int k, float **bucket, size_t *n, int *index, int *nodes, int cap,
                           int *count)
{
    if (!exec || k < 0 || !count || cap < 0 || (cap > 0 && !nodes)) return DVSOF_EINVAL;
    Exec *x = (Exec *)exec;
    const int nn = (int)x->nodes.size();
    std::vector<int> by_id(nn);
    for (int i = 0; i < nn; ++i) by_id[x->nodes[i].id] = i;
    // the k-th BUCKET mark in capture order
    std::vector<int> marks;
    for (int i = 0; i < nn; ++i)
        if (x->nodes[i].mark == DVSOF_MARK_BUCKET) marks.push_back(x->nodes[i].id);
    std::sort(marks.begin(), marks.end());
    if (k >= (int)marks.size()) return DVSOF_EINVAL;
    const XNode &m = x->nodes[by_id[marks[k]]];
    if (bucket) *bucket = m.mark_ptr;
    if (n) *n = m.mark_n;
    if (index) *index = m.mark_index;
    *count = (int)m.window.size();
    for (int j = 0; j < *count && j < cap; ++j) nodes[j] = by_id[m.window[j]];
    return DVSOF_OK;
}

int dvsof_exec_node_arg(void *exec, int i, int arg, size_t nbytes, void *out)
{
    if (!exec || !out) return DVSOF_EINVAL;
    Exec *x = (Exec *)exec;
    if (i < 0 || i >= (int)x->nodes.size() || arg < 0) return DVSOF_EINVAL;
    const XNode &n = x->nodes[i];
    if (!n.kernel || !n.kp.kernelParams || !n.kp.kernelParams[arg]) return DVSOF_EINVAL;
    memcpy(out, n.kp.kernelParams[arg], nbytes);
    return DVSOF_OK;
}

int dvsof_exec_create(void *graph_, void *const *side_streams, int n_side, void **out)
{
    if (!graph_ || !out || n_side < 0 || (n_side > 0 && !side_streams)) return DVSOF_EINVAL;
    hipGraph_t graph = (hipGraph_t)graph_;
    size_t nn = 0, ne = 0;
    DVSOF_HIP_TRY(hipGraphGetNodes(graph, nullptr, &nn));
    if (nn == 0) return DVSOF_EINVAL;
    std::vector<hipGraphNode_t> hn(nn);
    DVSOF_HIP_TRY(hipGraphGetNodes(graph, hn.data(), &nn));
    DVSOF_HIP_TRY(hipGraphGetEdges(graph, nullptr, nullptr, &ne));
    std::vector<hipGraphNode_t> ef(ne), et(ne);
    if (ne) DVSOF_HIP_TRY(hipGraphGetEdges(graph, ef.data(), et.data(), &ne));

    std::unordered_map<hipGraphNode_t, int> raw_index;
    for (size_t i = 0; i < nn; ++i) raw_index[hn[i]] = (int)i;
    std::vector<std::vector<int>> rdeps(nn), rsucc(nn);
    for (size_t e = 0; e < ne; ++e) {
        auto a = raw_index.find(ef[e]), b = raw_index.find(et[e]);
        if (a == raw_index.end() || b == raw_index.end()) return DVSOF_EINVAL;
        rdeps[b->second].push_back(a->second);
        rsucc[a->second].push_back(b->second);
    }
    // topological order; ties by position in the node list (= capture order of
    // a stream capture), so that the host issues kernels in the eager order
    std::vector<int> order, indeg(nn);
    order.reserve(nn);
    {
        std::vector<int> ready;
        for (size_t i = 0; i < nn; ++i) {
            indeg[i] = (int)rdeps[i].size();
            if (!indeg[i]) ready.push_back((int)i);
        }
        auto cmp = [](int a, int b) { return a > b; };   // min-heap on the list position
        std::make_heap(ready.begin(), ready.end(), cmp);
        while (!ready.empty()) {
            std::pop_heap(ready.begin(), ready.end(), cmp);
            const int i = ready.back();
            ready.pop_back();
            order.push_back(i);
            for (int s : rsucc[i])
                if (--indeg[s] == 0) {
                    ready.push_back(s);
                    std::push_heap(ready.begin(), ready.end(), cmp);
                }
        }
        if (order.size() != nn) return DVSOF_EINVAL;   // cycle: not a DAG
    }
    std::vector<int> pos(nn);
    for (size_t i = 0; i < nn; ++i) pos[order[i]] = (int)i;

    Exec *x = new Exec;
    x->nodes.resize(nn);
    const int max_lanes = 1 + n_side;
    x->max_lanes = max_lanes;
    std::vector<int> &tail = x->tail;
    for (size_t i = 0; i < nn; ++i) {
        XNode &n = x->nodes[i];
        n.ev = nullptr;
        const int r = order[i];
        hipGraphNodeType ty;
        hipError_t e = hipGraphNodeGetType(hn[r], &ty);
        if (e != hipSuccess) {
            destroy(x);
            return (int)e;
        }
        n.kernel = ty == hipGraphNodeTypeKernel;
        if (n.kernel) {
            e = hipGraphKernelNodeGetParams(hn[r], &n.kp);
            if (e != hipSuccess) {
                destroy(x);
                return (int)e;
            }
            if (!n.kp.func || (!n.kp.kernelParams && !n.kp.extra)) {
                destroy(x);
                return DVSOF_EINVAL;
            }
            if (n.kp.func == (void *)exec_mark_kernel && n.kp.kernelParams) {
                n.mark = *(const int *)n.kp.kernelParams[0];
                n.mark_index = *(const int *)n.kp.kernelParams[1];
                n.mark_ptr = *(float *const *)n.kp.kernelParams[2];
                n.mark_n = *(const size_t *)n.kp.kernelParams[3];
                ++x->n_marks;
            } else {
                ++x->n_kernels;
            }
        } else if (ty != hipGraphNodeTypeEmpty) {
            destroy(x);
            return DVSOF_EINVAL;      // memset / memcpy / host nodes: not a kernels-only step
        }
        // lane: extend the chain of a dependency that is still the tail of its
        // lane (lowest lane first); else open a lane; else share the last one
        std::vector<int> deps;
        for (int d : rdeps[r]) deps.push_back(pos[d]);
        std::sort(deps.begin(), deps.end());
        int lane = -1;
        for (int d : deps) {
            const int l = x->nodes[d].lane;
            if (tail[l] == d && (lane < 0 || l < lane)) lane = l;
        }
        if (lane < 0) {
            if ((int)tail.size() < max_lanes) {
                lane = (int)tail.size();
                tail.push_back(-1);
            } else {
                lane = max_lanes - 1;
            }
        }
        n.lane = n.lane0 = lane;
        n.id = (int)i;
        tail[lane] = (int)i;
        n.deps = n.deps_id = deps;
    }
    // The exchange window of every BUCKET mark, from the dependencies AS CAPTURED: the kernels
    // that follow the mark in the capture but neither its WAIT mark nor (without one) the JOIN
    // mark.  The rewrite below lets exactly these run beside the collective, which is only
    // right when none of them reads or writes the bucket -- dvsof_exec_mark_window hands the
    // set to the caller, who knows the kernels' argument layouts (capture.py refuses the
    // recording when one of them carries a pointer into the bucket).
    {
        std::vector<std::vector<int>> succ(nn);
        for (size_t i = 0; i < nn; ++i)
            for (int d : x->nodes[i].deps_id) succ[d].push_back((int)i);
        auto reach = [&](int from, std::vector<char> &seen) {
            std::vector<int> todo{from};
            seen[from] = 1;
            while (!todo.empty()) {
                const int v = todo.back();
                todo.pop_back();
                for (int s_ : succ[v])
                    if (!seen[s_]) {
                        seen[s_] = 1;
                        todo.push_back(s_);
                    }
            }
        };
        for (size_t m = 0; m < nn; ++m) {
            XNode &mk = x->nodes[m];
            if (mk.mark != DVSOF_MARK_BUCKET) continue;
            std::vector<char> behind(nn, 0), safe(nn, 0);
            reach((int)m, behind);
            bool has_wait = false;
            for (size_t j = 0; j < nn; ++j)
                if (x->nodes[j].mark == DVSOF_MARK_WAIT && x->nodes[j].mark_index == mk.mark_index) {
                    has_wait = true;
                    reach((int)j, safe);
                }
            if (!has_wait)
                for (size_t j = 0; j < nn; ++j)
                    if (x->nodes[j].mark == DVSOF_MARK_JOIN && behind[j]) reach((int)j, safe);
            for (size_t j = 0; j < nn; ++j)
                if (j != m && behind[j] && !safe[j] && x->nodes[j].kernel && !x->nodes[j].mark)
                    mk.window.push_back((int)j);
        }
    }
    // A BUCKET mark is a side effect of its stream, not a producer: what follows it in the
    // capture follows what PRECEDED it.  Kernels inherit the mark's dependencies instead of
    // depending on the mark (else the kernel behind a mark is tied to the mark's lane and, in
    // another lane, pays a cross-lane wait per bucket); the other marks keep theirs (the JOIN
    // has to be issued behind every collective, a WAIT behind its own).
    for (size_t i = 0; i < nn; ++i) {
        XNode &n = x->nodes[i];
        if (n.mark) continue;
        std::vector<int> out;
        std::vector<int> todo(n.deps_id.begin(), n.deps_id.end());
        while (!todo.empty()) {
            const int d = todo.back();
            todo.pop_back();
            if (x->nodes[d].mark == DVSOF_MARK_BUCKET) {
                for (int dd : x->nodes[d].deps_id) todo.push_back(dd);
            } else if (std::find(out.begin(), out.end(), d) == out.end()) {
                out.push_back(d);
            }
        }
        std::sort(out.begin(), out.end());
        n.deps = n.deps_id = out;
    }
    // Nodes of the exchange stream: a kernel whose every dependency is a WAIT mark or such a kernel
    // (parallel.GradReducer captures a bucket's optimizer update on the exchange stream, behind the
    // bucket's WAIT mark: the update follows the collective in stream order and no compute lane waits
    // for either), and the WAIT marks all of whose successors are such kernels.
    {
        std::vector<std::vector<int>> succ(nn);
        for (size_t i = 0; i < nn; ++i)
            for (int d : x->nodes[i].deps_id) succ[d].push_back((int)i);
        for (size_t i = 0; i < nn; ++i) {       // capture order is a topological order
            XNode &n = x->nodes[i];
            if (n.mark || !n.kernel || n.deps_id.empty()) continue;
            bool all = true;
            for (int d : n.deps_id)
                if (!(x->nodes[d].mark == DVSOF_MARK_WAIT || x->nodes[d].xlane)) all = false;
            n.xlane = all;
        }
        for (size_t k = nn; k-- > 0;) {     // (descending: a WAIT mark's successor may be the next WAIT mark
            XNode &n = x->nodes[k];         // of the branch -- an update that collects several buckets)
            if (n.mark != DVSOF_MARK_WAIT || succ[k].empty()) continue;
            bool all = true;
            for (int s_ : succ[k])
                if (!(x->nodes[s_].xlane || x->nodes[s_].mark == DVSOF_MARK_JOIN)) all = false;
            n.xlane = all;
        }
        for (auto &n : x->nodes) n.lane_plan = n.lane;
    }
    for (int l = 0; l < n_side; ++l) x->side.push_back((hipStream_t)side_streams[l]);
    {
        const int rc = wire(x);
        if (rc) {
            destroy(x);
            return rc;
        }
    }
    *out = x;
    return DVSOF_OK;
}

int dvsof_exec_info(void *exec, int *n_kernels, int *n_lanes, int *n_events, int *n_waits,
                    int *lane_kernels, int max_lanes)
{
    if (!exec) return DVSOF_EINVAL;
    Exec *x = (Exec *)exec;
    if (n_kernels) *n_kernels = x->n_kernels;
    if (n_lanes) *n_lanes = (int)x->tail.size();
    if (n_events) *n_events = x->n_events;
    if (n_waits) *n_waits = x->n_waits;
    if (lane_kernels) {
        for (int l = 0; l < max_lanes; ++l) lane_kernels[l] = 0;
        for (auto &n : x->nodes)
            if (n.kernel && !n.mark && n.lane < max_lanes) ++lane_kernels[n.lane];
    }
    return DVSOF_OK;
}

int dvsof_exec_node(void *exec, int i, int *lane, float *us, int *n_waits, char *name, int name_len)
{
    if (!exec) return DVSOF_EINVAL;
    Exec *x = (Exec *)exec;
    if (i < 0 || i >= (int)x->nodes.size()) return DVSOF_EINVAL;
    const XNode &n = x->nodes[i];
    if (lane) *lane = n.lane;
    if (us) *us = n.us;
    if (n_waits) *n_waits = (int)n.wait.size();
    if (name && name_len > 0) {
        const char *s = n.kernel && n.kp.kernelParams ? hipKernelNameRefByPtr(n.kp.func, nullptr) : nullptr;
        if (!s) s = n.kernel ? "?" : "(empty)";
        int k = 0;
        for (; k < name_len - 1 && s[k]; ++k) name[k] = s[k];
        name[k] = 0;
    }
    return DVSOF_OK;
}

int dvsof_exec_calibrate(void *exec, void *stream)
{
    if (!exec) return DVSOF_EINVAL;
    Exec *x = (Exec *)exec;
    hipStream_t st = as_stream(stream);
    const int nn = (int)x->nodes.size();
    std::vector<hipEvent_t> ev(nn + 1, nullptr);
    int rc = DVSOF_OK;
    auto run = [&]() -> int {
        for (auto &e : ev) DVSOF_HIP_TRY(hipEventCreate(&e));
        DVSOF_HIP_TRY(hipEventRecord(ev[0], st));
        for (int i = 0; i < nn; ++i) {
            XNode &n = x->nodes[i];
            if (n.mark) {
                // the exchange in stream order on the one stream of this pass
                if (x->comm && n.mark == DVSOF_MARK_BUCKET) {
                    const int rc_ = dvsof_allreduce_bucket(x->comm, n.mark_ptr, n.mark_n, (void *)st);
                    if (rc_) return rc_;
                }
            } else if (n.kernel) {
                const int rc_ = launch_node(n, st);
                if (rc_) return rc_;
            }
            DVSOF_HIP_TRY(hipEventRecord(ev[i + 1], st));
        }
        DVSOF_HIP_TRY(hipStreamSynchronize(st));
        for (int i = 0; i < nn; ++i) {
            float ms = 0.f;
            DVSOF_HIP_TRY(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            x->nodes[i].us = x->nodes[i].mark ? 0.f : x->nodes[i].kernel ? std::max(ms * 1e3f, 0.5f) : 0.f;
        }
        return DVSOF_OK;
    };
    rc = run();
    for (auto e : ev)
        if (e) (void)hipEventDestroy(e);
    if (rc) return rc;
    // DVSOF_EXEC_PLAN = paths | list | chain: that plan; default: every plan is tried on
    // the next steps (real steps, timed on the device), the fastest stays
    static const char *want = getenv("DVSOF_EXEC_PLAN");
    static const double hop = getenv("DVSOF_EXEC_HOP") ? atof(getenv("DVSOF_EXEC_HOP")) : 8.0;
    x->cands.clear();
    x->trial_next = x->trial_cur = -1;
    if (x->max_lanes < 2) return wire(x);
    x->cands.push_back(plan_by_time(x));
    x->cands.push_back(plan_by_list(x, hop));
    x->cands.push_back(plan_chain(x));
    int fixed = -1;
    for (size_t i = 0; want && i < x->cands.size(); ++i)
        if (!strcmp(want, x->cands[i].name)) fixed = (int)i;
    // Under an exchange every rank takes the SAME plan, the capture's own two-stream split
    // ("chain": what the trials picked in every 1-rank and loopback run): timed trials would
    // measure the other ranks' arrival at the collectives as much as this rank's kernels, ranks
    // could settle on different plans, and the trial steps themselves (host waits for a step's
    // end) are one more place where ranks wait for each other.  DVSOF_EXEC_PLAN overrides.
    if (fixed < 0 && !want && x->comm)
        for (size_t i = 0; i < x->cands.size(); ++i)
            if (!strcmp("chain", x->cands[i].name)) fixed = (int)i;
    if (fixed >= 0) {
        apply(x, x->cands[fixed]);
        x->cands.clear();
        return wire(x);
    }
    if (!x->t0) DVSOF_HIP_TRY(hipEventCreate(&x->t0));
    if (!x->t1) DVSOF_HIP_TRY(hipEventCreate(&x->t1));
    x->trial_next = 0;
    apply(x, x->cands[0]);
    return wire(x);
}

// Steps after calibration: plan k % n on trial k, `rounds` rounds; then the fastest.
static int next_trial(Exec *x)
{
    static const int rounds = getenv("DVSOF_EXEC_TRIALS") ? std::max(1, atoi(getenv("DVSOF_EXEC_TRIALS"))) : 2;
    const int n = (int)x->cands.size();
    if (x->trial_cur >= 0) {    // the step in flight was a trial: its device time
        DVSOF_HIP_TRY(hipEventSynchronize(x->t1));
        float ms = 0.f;
        DVSOF_HIP_TRY(hipEventElapsedTime(&ms, x->t0, x->t1));
        Plan &p = x->cands[x->trial_cur];
        p.best_us = std::min(p.best_us, ms * 1e3f);
        x->trial_cur = -1;
    }
    if (x->trial_next < n * rounds) {
        x->trial_cur = x->trial_next++ % n;
        apply(x, x->cands[x->trial_cur]);
        return wire(x);
    }
    int best = 0;
    for (int i = 1; i < n; ++i)
        if (x->cands[i].best_us < x->cands[best].best_us) best = i;
    apply(x, x->cands[best]);
    x->trial_next = -1;
    return wire(x);
}

int dvsof_exec_launch(void *exec, void *stream)
{
    if (!exec) return DVSOF_EINVAL;
    Exec *x = (Exec *)exec;
    hipStream_t main = as_stream(stream);
    if (x->trial_next >= 0) {
        const int rc_ = next_trial(x);
        if (rc_) return rc_;
    }
    const int L = (int)x->tail.size();
    auto lane_stream = [&](int l) { return l == 0 ? main : l <= (int)x->side.size() ? x->side[l - 1] : x->ustream; };
    if (x->trial_cur >= 0) DVSOF_HIP_TRY(hipEventRecord(x->t0, main));
    if (L > 1) {   // the side lanes start behind whatever precedes the step on `stream`
        DVSOF_HIP_TRY(hipEventRecord(x->fork, main));
        for (int l = 1; l < L; ++l) DVSOF_HIP_TRY(hipStreamWaitEvent(lane_stream(l), x->fork, 0));
    }
    for (auto &n : x->nodes) {
        hipStream_t st = lane_stream(n.lane);
        for (int d : n.wait) DVSOF_HIP_TRY(hipStreamWaitEvent(st, x->nodes[d].ev, 0));
        if (n.mark) {
            const int rc_ = run_mark(x, n, st);
            if (rc_) return rc_;
        } else if (n.kernel) {
            const int rc_ = launch_node(n, st);
            if (rc_) return rc_;
        }
        if (n.ev) DVSOF_HIP_TRY(hipEventRecord(n.ev, st));
    }
    for (int l = 1; l < L; ++l) {   // `stream` continues behind every lane
        DVSOF_HIP_TRY(hipEventRecord(x->join[l - 1], lane_stream(l)));
        DVSOF_HIP_TRY(hipStreamWaitEvent(main, x->join[l - 1], 0));
    }
    if (x->trial_cur >= 0) DVSOF_HIP_TRY(hipEventRecord(x->t1, main));
    return DVSOF_OK;
}

int dvsof_exec_plan(void *exec, char *name, int name_len, int *settled)
{
    if (!exec) return DVSOF_EINVAL;
    Exec *x = (Exec *)exec;
    if (settled) *settled = x->trial_next < 0;
    if (name && name_len > 0) {
        int k = 0;
        for (; k < name_len - 1 && x->plan_name[k]; ++k) name[k] = x->plan_name[k];
        name[k] = 0;
    }
    return DVSOF_OK;
}

int dvsof_exec_destroy(void *exec)
{
    if (!exec) return DVSOF_EINVAL;
    destroy((Exec *)exec);
    return DVSOF_OK;
}

}  // extern "C"
