// sca_hip.hip -- host side of libsca_hip.so: context, HBM layout, launches, C-ABI (include/sca_hip.h).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>          // types and prototypes only: the library is loaded with dlopen when sca_comm_init is called
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sca_hip.h"
#include "sca_kernels.hip.h"
#include "sca_kdbuild.hip.h"
#include "sca_grid.hip.h"
#include "sca_dubins.hpp"
#include "sca_tracker.hip.h"
#include "sca_partition.hip.h"

using namespace sca;

namespace {

// grid cap for the wave-per-agent kernels (they grid-stride): 256 CUs x 8 workgroups
constexpr int MAX_GRID = 1 << 30;      // one agent per wavefront: measured faster than grid-striding (k_solve keeps 2 waves/SIMD)

// libm calls that must not be folded by the compiler (the candidate table follows the reference's
// Python expression literally: z ** 2 is pow(z, 2), not z * z)
double (*volatile p_pow)(double, double) = std::pow;
double (*volatile p_acos)(double) = std::acos;

// Replica of KDTree.buildAgentTreeRecursive / buildObstacleTreeRecursive (kdTree.py:60-122,162-227),
// iterative with an explicit work list.  Children live at node+1 and node+2*leftSize.
void kd_build_host(int n, const double *pos, int32_t *ids, std::vector<KdNode> &tree) {
    tree.assign(n > 0 ? (size_t)(2 * n) : 1, KdNode{});
    if (n <= 0) return;
    struct Job { int begin, end, node; };
    std::vector<Job> work;
    work.push_back({0, n, 0});
    while (!work.empty()) {
        const Job jb = work.back();
        work.pop_back();
        KdNode &nd = tree[jb.node];
        nd.begin = jb.begin; nd.end = jb.end; nd.left = 0; nd.right = 0;
        for (int k = 0; k < 3; k++) nd.mn[k] = nd.mx[k] = pos[3 * ids[jb.begin] + k];
        for (int i = jb.begin + 1; i < jb.end; i++)
            for (int k = 0; k < 3; k++) {
                const double v = pos[3 * ids[i] + k];
                if (v > nd.mx[k]) nd.mx[k] = v;
                if (v < nd.mn[k]) nd.mn[k] = v;
            }
        if (jb.end - jb.begin > MAX_LEAF) {
            const double d0 = nd.mx[0] - nd.mn[0], d1 = nd.mx[1] - nd.mn[1], d2 = nd.mx[2] - nd.mn[2];
            const int c = (d0 > d1 && d0 > d2) ? 0 : (d1 > d2 ? 1 : 2);
            const double split = 0.5 * (nd.mx[c] + nd.mn[c]);
            int lo = jb.begin, hi = jb.end;
            while (lo < hi) {
                while (lo < hi && pos[3 * ids[lo] + c] < split) lo++;
                while (hi > lo && pos[3 * ids[hi - 1] + c] >= split) hi--;
                if (lo < hi) { std::swap(ids[lo], ids[hi - 1]); lo++; hi--; }
            }
            int leftSize = lo - jb.begin;
            if (leftSize == 0) { leftSize = 1; lo++; }
            nd.left = jb.node + 1;
            nd.right = jb.node + 2 * leftSize;
            // order of construction does not matter: the two sub-ranges are disjoint
            work.push_back({lo, jb.end, nd.right});
            work.push_back({jb.begin, lo, nd.left});
        }
    }
}

// query view of a host-built tree: node headers + both children's boxes (KdWide)
void kd_widen_host(int n, const std::vector<KdNode> &tree, std::vector<KdWide> &wide) {
    wide.assign(n > 0 ? (size_t)(2 * n) : 1, KdWide{});
    if (n <= 0) return;
    std::vector<int> st{0};
    while (!st.empty()) {
        const int i = st.back();
        st.pop_back();
        const KdNode &nd = tree[i];
        KdWide &w = wide[i];
        w.begin = nd.begin; w.end = nd.end; w.left = nd.left; w.right = nd.right;
        if (nd.end - nd.begin > MAX_LEAF) {
            for (int k = 0; k < 3; k++) {
                w.bx[kdw_idx(0, 0, k)] = tree[nd.left].mn[k]; w.bx[kdw_idx(0, 1, k)] = tree[nd.left].mx[k];
                w.bx[kdw_idx(1, 0, k)] = tree[nd.right].mn[k]; w.bx[kdw_idx(1, 1, k)] = tree[nd.right].mx[k];
            }
            st.push_back(nd.left); st.push_back(nd.right);
        }
    }
}

void candidate_table_host(int num_N, double *unit, double *phi) {
    const double param_phi = (std::sqrt(5.0) - 1.0) / 2.0;                         // scaPolicy.py:191
    for (int n = 1; n <= num_N; n++) {
        const double z_n = (double)(2 * n - 1) / num_N - 1;                        // :197
        const double c = std::sqrt(1 - p_pow(z_n, 2.0));
        const double ang = 2 * M_PI * n * param_phi;
        const double x_n = c * std::cos(ang), y_n = c * std::sin(ang);             // :198-199
        unit[n - 1] = x_n; unit[num_N + n - 1] = y_n; unit[2 * num_N + n - 1] = z_n;
        // get_phi (util.py:145) of the direction; rad in {0.5, 1.0} scales both atan2 arguments exactly
        double ph = (y_n >= 0) ? std::atan2(y_n, x_n) : 2 * M_PI + std::atan2(y_n, x_n);
        double t = std::trunc(ph * EPS5);
        if (t == 0.0) t = 0.0;
        phi[n - 1] = t;
    }
}

// smallest double c with acos(c) <= mhc under the host libm: `theta <= max_heading_change` (util.py:16-17)
// becomes `costheta >= thr` exactly (acos is monotone).
double cos_threshold(double mhc) {
    if (p_acos(-1.0) <= mhc) return -1.0;
    if (!(p_acos(1.0) <= mhc)) return 2.0;      // nothing satisfies
    double lo = -1.0, hi = 1.0;                 // acos(lo) > mhc, acos(hi) <= mhc
    while (std::nextafter(lo, 2.0) < hi) {
        const double mid = lo + (hi - lo) * 0.5;
        if (mid <= lo || mid >= hi) break;
        if (p_acos(mid) <= mhc) hi = mid; else lo = mid;
    }
    return hi;
}

}  // namespace

// RCCL entry points, resolved at run time (a box without RCCL still loads libsca_hip.so; single-GPU use never touches it)
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static const char *rccl_load() {
    if (g_rccl.handle) return nullptr;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return "librccl.so not found (dlopen)";
    RcclApi a;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString) return "librccl.so lacks an expected symbol";
    a.handle = h;
    g_rccl = a;
    return nullptr;
}

struct sca_ctx {
    int device = 0;
    int max_n = 0, max_m = 0, n = 0, m = 0;
    Params P{};
    DeviceView d{};
    PubRec *rec_own = nullptr, *rec_new_own = nullptr;
    hipStream_t stream_own = nullptr, stream = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // policy pass: 0..3, env update: 4, 5
    float ms_nbr = 0, ms_solve = 0, ms_update = 0;
    // per-launch profiling of sca_run_steps: event pairs around every k_neighbors_kd / k_solve launch
    bool profiling = false;
    std::vector<hipEvent_t> pool;       // 4 events per profiled pass: around K1 (on its stream), around k_solve
    int pool_used = 0;
    std::vector<hipEvent_t> pool_kd;    // 2 events around every 16th kd build while profiling (sca_last_kd_build_ms)
    int pool_kd_used = 0;
    float ms_kd_build = 0.0f;
    std::vector<hipEvent_t> pool_xch;   // 2 events per profiled step around the in-library ncclAllGather (sca_last_exchange_ms)
    int pool_xch_used = 0;
    float ms_exchange = 0.0f;
    std::vector<hipEvent_t> pool_trk;   // 2 events per step on the stream the re-plan kernels run on: before / after them
    int pool_trk_used = 0;
    float ms_replan = 0;
    // K3: the ORCA3D-LP agents, ascending ids (static: sca_set_agents), solved one lane per agent by k_lp
    int32_t *lp_list = nullptr;
    std::vector<int32_t> h_lp_list;
    bool shard_emulation = false;       // measurement aid: a partial shard without a communicator, the others stand still
    std::vector<double> h_pos;          // host mirror of positions for the kd build
    bool h_pos_valid = false;
    std::vector<int32_t> h_perm;
    std::vector<KdNode> h_tree;
    std::vector<PubRec> h_rec;
    KdScratch kd{};
    GridDev grid{};                     // SCA_NBR_GRID (sca_grid.hip.h)
    int nbr_mode = SCA_NBR_KDTREE;      // neighbour structure of the last policy pass: K4's fallback looks there
    // depth of the large-node part of the tree, read back asynchronously (never waited for) to size the next build
    int *kd_host_counts = nullptr;      // pinned
    hipEvent_t kd_ev = nullptr;
    bool kd_ev_pending = false;
    unsigned kd_builds = 0;             // device builds so far: the level statistics are read back every 8th
    unsigned kd_gen = 0, kd_ev_gen = 0; // positions replaced from outside (sca_set_state ...): statistics of older trees do not apply
    bool kd_nohint = false;             // SCA_KD_NOHINT=1: ignore the statistics of earlier builds (diagnostics)
    int action_fb_max = 0;              // shards up to this many agents run the fallback sweep inside the epilogue's launch (k_action_fb; SCA_ACTION_FB_MAX)
    int solve_fb_max = 0;               // shards up to this many agents solve and fall back in one launch (SCA_SOLVE_FB_MAX; default: two wavefronts per SIMD)
    bool kd_top = true;                 // SCA_KD_TOP=0: trees of <= KT_M members through the level passes as well (tests, measurements)
    int kd_tail_level = -1;             // SCA_KD_TAIL_LEVEL=l: the level at which the tail launch takes over (tuning / tests; -1: from the statistics)
    int kd_single_hint = 0;             // 1 + first level whose nodes all fit one chunk in an earlier build (0: unknown)
    unsigned kd_token = 0;              // launch token of the chained scan (never reused)
    int solve_split = -1;               // k_solve in two launches (k_solve_sweep beside the re-plans, k_solve_pick4 behind them): -1 when the
                                        // tracker is overlapped, 0 never, 1 always (SCA_SOLVE_SPLIT; the parity tests run both)
    int k1_force = -1;                  // K1 variant: -1 choose by shard size, 0 one agent per wavefront, 1 four (k_neighbors_kd4)
    bool perm_on_device = false;        // the live agentIDs permutation is d.aperm (device build) rather than h_perm
    bool agents_set = false, state_set = false;
    bool state_fresh = false;           // flags came from sca_set_state: the next env update checks arrived agents against obstacles once
    bool near_valid = false;            // K1's collision-candidate lists describe the current records
    double max_radius = 0, max_obs_radius = 0, max_pref_speed = 0;
    std::string err;
    double *tab = nullptr;
    // device-side v_pref tracker (sca_tracker.hip.h)
    TrackDev trk{};
    sca_dubins::TrackView trk_view{};
    double *trk_goal_heading = nullptr;
    bool trk_on = false, trk_in_pass = false;
    // the re-plans run on a stream of their own, next to the kd build and the neighbour query of the same pass
    hipStream_t trk_stream = nullptr;   // round 2: the re-plans are the critical path and stay on the main stream; the neighbour
                                        // structure (K0) and the neighbour query (K1) of the pass run on this one beside them
    hipEvent_t trk_fork = nullptr, trk_join = nullptr;
    hipStream_t nbr_stream = nullptr;   // where K0 / K1 of the current pass go: trk_stream when overlapped, else the main stream
    bool kd_ahead_enqueue = false;      // (timeline builds: the build being enqueued belongs to the next pass)
    Params P_ctx{};                     // sca_create's parameters; P holds the ENVELOPE of the agents' own while sca_set_agent_params is in force
    AgentPar *ap_dev = nullptr;         // [n] per-agent solver attributes on the device (DeviceView::ap), null: one value per context
    double *ap_nd = nullptr;            // [n] the agents' neighborDist alone, for the tracker (TrackView::nd_per_agent)
    int32_t *h_done = nullptr;          // pinned: K4's 256 counters (stride 32) + the kd build's error word (read_active)
    std::vector<std::array<double, 3>> trk_classes;   // (turning radius, pitch_lo, pitch_hi) of every class of tracked agents; empty: one class = the view's scalars
    double *trk_R_pa = nullptr;         // [n] agent.turning_radius (device), for the decision
    uint8_t *trk_cls = nullptr;         // [n] the agent's class (device)
    double *trk_plo_pa = nullptr, *trk_phi_pa = nullptr;   // [n] agent.pitchlims (device): the per-agent form (more than TRK_MAX_CLASSES classes)
    int auto_div = 8;                   // SCA_NBR_AUTO backs off to the plain kd pass (for 256 passes) once the grid query lists more than 1 / auto_div of the
                                        // shard for the kd query (SCA_AUTO_BACKOFF_DIV: measurements)
    bool trk_many = false;              // the per-agent form: every re-plan by a wavefront of its own, which reads ITS agent's three values
    double trk_enable_vals[3] = {1.5, 0, 0};   // (turning radius, pitch_lo, pitch_hi) of sca_device_tracker_enable: what "back to one value" restores
    int *trk_host_count = nullptr;      // pinned: the re-plan count of an earlier pass, copied back without ever being waited for
    hipEvent_t trk_count_ev = nullptr;
    bool trk_count_pending = false;
    int trk_last_count = -1;            // -1: unknown
    unsigned trk_passes = 0;
    int forms = 0;                      // SCA_FORM_* of the last policy pass
    // cell-owner partition of SCA_NBR_GRID (sca_partition.hip.h)
    PartDev part{};
    bool part_on = false;
    int part_rank = 0, part_nranks = 1;
    int part_counts[8] = {0};            // [0] owned, [1] halo: UPPER BOUNDS the launches are sized with (the kernels read the exact
                                        // counts on the device); [4] error bits as last seen
    int *part_host = nullptr;            // pinned: the counts of an earlier commit, copied back without being waited for
    hipEvent_t part_ev = nullptr;
    bool part_pending = false;
    int part_known[2] = {0, 0};          // the last exact counts the host has seen, and how many commits ago
    int part_age = 0, part_copy_age = 0;
    uint8_t *part_scratch[2] = {nullptr, nullptr};   // outgoing messages of a rank that runs alone (sca_run_steps: one rank, or emulation)
    int cus = 256, simds = 1024;         // the device's compute units / SIMDs (hipDeviceProp): every launch heuristic below is stated in
                                        // wavefronts per SIMD and scaled with them; the figures were measured on a 256-CU MI355X
    int kd_rank_capacity = 1 << 30;     // workgroups of k_kd_lv_rank the device holds at once (occupancy x CUs)
    bool kd_force_ticket = false;
    // SCA_NBR_AUTO: the grid query for everybody, the kd query for the agents it lists; the kd BUILD (the permutation is history: it
    // runs every step) on a stream of its own beside the grid build and query
    hipStream_t kd_stream = nullptr;
    hipEvent_t ev_auto_fork = nullptr, ev_auto_k1g = nullptr, ev_auto_kd = nullptr, ev_auto_moved = nullptr, ev_auto_cnt = nullptr;
    int32_t *kdq_list = nullptr, *kdq_count = nullptr;
    int *auto_ticket = nullptr;         // k_neighbors_kd_auto's last-workgroup ticket
    unsigned *auto_busy = nullptr;      // device word, bit 0: somebody is listed for the kd query and it has not answered yet (hipStreamWaitValue32)
    unsigned auto_seq = 0;
    bool auto_waitvalue = true;         // SCA_AUTO_EVENT_WAIT=1: an event wait behind the kd query instead (the build is then on every pass's path)
    hipEvent_t ev_auto_kdq[4] = {nullptr, nullptr, nullptr, nullptr};   // [seq & 3] behind the kd query of pass seq (a launch on kd_stream) or, in the
                                                         // launch-free form, behind its build's last kernel; the pass two on reuses the list: [(seq - 2) & 3]
    unsigned *auto_sync = nullptr;      // device words of the launch-free form (KdTail): [0] k_kd_block's ticket, [1] the last pass whose tree is complete, [2] the grid query's ticket
    bool auto_tail_ok = false;          // the tail form is available (with the wait-value form of the pass's wait; SCA_AUTO_NO_TAIL=1, read at sca_create: never)
    bool auto_no_tail = false;
    int auto_tail_max = 0;              // ... and taken while the list lengths that come back stay at or below this (SCA_AUTO_TAIL_MAX).  0: while NOBODY is
                                        // listed -- one workgroup answering even a handful of agents per pass lost against the launch form over a
                                        // whole c3 episode (3000 steps: 0.172 ms per step at 32, 0.121-0.125 at 8, 0.099-0.101 at 0, 0.104 launch form)
    unsigned kd_tail_seq = 0;           // the pass whose build was enqueued in the tail form (0: none)
    bool kdq_on_kd_stream[4] = {false, false, false, false};   // [seq & 3]: pass seq's kd query was a launch on kd_stream (its list's next user must wait for it)
    KdTail kd_tail_arg = {};            // what the build being enqueued hands its k_kd_block (seq = 0 outside an AUTO build)
    hipEvent_t kd_block_stop = nullptr; // ... and the event to record behind it
    hipEvent_t ev_auto_gather[2] = {nullptr, nullptr};   // the gather kernel of the last two builds: the integrate stage must not write into
    unsigned auto_builds = 0;                            // the record buffer a build's gather still reads (the two buffers alternate)
    bool lazy_join = false;             // sca_env_step left kd_stream unjoined: the next entry point other than sca_env_step joins (API_ENTER)
    bool auto_unjoined = false;         // kd_stream may still be working on the last AUTO pass's tree / kd query
    int *kdq_host = nullptr;            // pinned: the list length of an earlier pass (never waited for)
    bool kdq_pending = false;
    int kdq_last = -1;                  // -1: unknown
    int auto_backoff = 0;               // passes left to run as plain SCA_NBR_KDTREE (the grid listed too many agents: ties everywhere)
    bool auto_ran = false, auto_fits = false;   // the last pass was an AUTO pass / the grid's candidate lists cover the collision reach
    bool kd_ahead = false;              // the kd build of the NEXT pass is already enqueued on kd_stream (sca_run_steps, behind the integrate stage)
    unsigned auto_passes = 0;
    bool ext_stop = true;               // the fork / join / hand-over events as stop events of the kernels they follow (LAUNCH_REC); SCA_EXT_STOP=0: records
    hipEvent_t action_stop = nullptr;   // sca_run_steps: the event to record behind this pass's k_action (the moved positions), if any
    hipEvent_t finish_stop = nullptr;   // sca_run_steps: the event to record behind the step's last kernel (the next pass's fork), if any
    bool fork_ready = false;            // ... it was: the pass that follows at once does not record its fork
    int kd_wave_cap = 0;                // largest subtree handed to k_kd_block (SCA_KD_WAVE_CAP: 256 .. 1536); 0: chosen per pass
    bool trk_group_fuse = true;         // k_track_group (decision + 64-lane search in one launch) for shards of <= TRK_SPEC4_MAX agents; SCA_TRACKER_NOGROUPFUSE=1: never
    bool trk_fuse = true;               // k_track_replan allowed (SCA_TRACKER_NOFUSE switches it off: A/B measurements, tests)
    unsigned prof_tick = 0;             // with profiling on, every 16th pass carries the event pairs (six records, ~35 us on that pass)
    bool trk_serial = false;            // SCA_TRACKER_SERIAL=1: everything on one stream (diagnostics)
    bool trk_quad = true;               // SCA_TRACKER_NOQUAD=1: lane-per-plan kernel only (diagnostics)
    // multi-GPU exchange inside the library (sca_comm_init): one ncclAllGather of the shard's moved records per step
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_nranks = 1;
};

#define CHK(ctx, call)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return SCA_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)
#define ARG(ctx, cond)                                                                         \
    do {                                                                                       \
        if (!(cond)) { (ctx)->err = std::string("bad argument: ") + #cond; return SCA_ERR_ARG; } \
    } while (0)

// a kernel and "record `ev` behind it": the event as the dispatch's own completion signal (hipExtLaunchKernelGGL's stopEvent) -- one packet in
// the queue instead of a dispatch and a barrier packet
#define LAUNCH_REC(ctx, ev, kern, grid, block, stream, ...)                                                            \
    do {                                                                                                               \
        if ((ctx)->ext_stop) hipExtLaunchKernelGGL(kern, grid, block, 0, stream, nullptr, ev, 0, __VA_ARGS__);         \
        else { hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__); CHK(ctx, hipEventRecord(ev, stream)); }  \
    } while (0)
// ... with an event only if there is one (`ev` may be null: a plain launch)
#define LAUNCH_OPT(ctx, ev, kern, grid, block, stream, ...)                                                            \
    do {                                                                                                               \
        if ((ev) != nullptr) LAUNCH_REC(ctx, ev, kern, grid, block, stream, __VA_ARGS__);                              \
        else hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__);                                            \
    } while (0)

template <class T>
static int dalloc(sca_ctx *c, T **p, size_t count) {
    CHK(c, hipMalloc((void **)p, sizeof(T) * (count ? count : 1)));
    CHK(c, hipMemsetAsync(*p, 0, sizeof(T) * (count ? count : 1), c->stream));
    return 0;
}

// sca_env_step (the drop-in loop's one call per step) returns without putting kd_stream in front of the context's stream -- an SCA_NBR_AUTO
// pass leaves the tree build and the kd query of the listed agents there, and the next pass copes with that by itself, as inside a burst of
// sca_run_steps.  Every OTHER entry point joins first: it may read the tree, the permutation or the lists, or enqueue work that does.
static int auto_join(sca_ctx *c);
static int api_enter(sca_ctx *c);
#define API_ENTER(ctx)                                                                          \
    do {                                                                                       \
        if (!(ctx)) return SCA_ERR_ARG;                                                        \
        if (int r_ = api_enter(ctx)) return r_;                                                \
    } while (0)

extern "C" {

// A version-100 caller's sca_params is 56 bytes (no dt_nominal) with `reserved` = 0 where struct_bytes now sits: sca_default_params keeps
// writing exactly those 56 bytes, sca_create reads dt_nominal only from a struct that says it has one.
void sca_default_params_v2(sca_params *p, int32_t struct_bytes) {
    p->neighbor_dist = 10.0; p->time_step = 0.1; p->time_horizon = 10.0; p->max_speed = 1.0;
    p->max_heading_change = M_PI / 4; p->near_goal_threshold = 0.5; p->max_neighbors = 16;
    const bool has_dt = struct_bytes >= (int32_t)(offsetof(sca_params, dt_nominal) + sizeof(double));
    p->struct_bytes = has_dt ? struct_bytes : 0;
    if (has_dt) p->dt_nominal = 0.1;
}
void sca_default_params(sca_params *p) { sca_default_params_v2(p, 0); }
int sca_version(void) { return 102; }   // 102: sca_params.struct_bytes + sca_default_params_v2, selftest codes 11-14, SCA_FORM_ACTION_FB / _AUTO_TAIL

const char *sca_last_error(const sca_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int sca_candidate_table(int num_N, double *unit, double *phi_num) {
    if (num_N <= 0 || !unit || !phi_num) return SCA_ERR_ARG;
    candidate_table_host(num_N, unit, phi_num);
    return 0;
}

int sca_kd_build_host(int n, const double *pos, int32_t *perm, double *tree_out) {
    if (n < 0 || (n > 0 && (!pos || !perm))) return SCA_ERR_ARG;
    std::vector<KdNode> tree;
    kd_build_host(n, pos, perm, tree);
    if (tree_out)
        for (int i = 0; i < 2 * n - 1; i++) {
            double *t = tree_out + 10 * (size_t)i;
            t[0] = tree[i].begin; t[1] = tree[i].end; t[2] = tree[i].left; t[3] = tree[i].right;
            for (int k = 0; k < 3; k++) { t[4 + k] = tree[i].mn[k]; t[7 + k] = tree[i].mx[k]; }
        }
    return 0;
}

// ---- what the tracker's parity claim is conditional on ----------------------------------------------------------------------
// The tracker (host and device) computes sin / cos / atan2 / acos / x ** 2 with a restatement of ONE libm build: GNU C Library
// 2.35, x86-64, the FMA variants its ifunc resolvers pick on AVX2 machines (sca_glibc_math.h) -- the libm the golden fixtures were
// recorded with.  A Python reference run on THIS host calls THIS host's libm; where that is another build (another glibc, a CPU
// without FMA, aarch64) the reference itself would print other last bits than the fixtures, and the library would keep printing
// 2.35's.  sca_libm_check compares the two on a fixed argument set (5 x 4096 points over the ranges a flight path produces) so
// that the condition is visible at run time: 0 = this host's libm gives the restated bits, 1 = it does not (mismatch counts per
// function in mismatches[5]: sin, cos, atan2, acos, pow(x, 2)).  Informational: nothing in the library changes its behaviour.
static double (*volatile p_sin)(double) = std::sin;
static double (*volatile p_cos)(double) = std::cos;
static double (*volatile p_atan2)(double, double) = std::atan2;
static int g_libm_state = -1;                   // -1 unknown, 0 equal, 1 different
static int64_t g_libm_bad[5] = {0, 0, 0, 0, 0};
static std::once_flag g_libm_once;              // trackers may be created from several threads: one check, no race on the statics
static void libm_check_once();
static int libm_check_run() {
    std::call_once(g_libm_once, libm_check_once);
    return g_libm_state;
}
static void libm_check_once() {
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    auto u01 = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };
    auto same = [](double a, double b) { return std::memcmp(&a, &b, sizeof(double)) == 0 || (a != a && b != b); };
    int64_t bad[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4096; i++) {
        const int k = i & 7;
        // angles: fractions of a turn, many turns (path angles are un-reduced sums), tiny, next to multiples of pi / 2
        double a = k < 3 ? (u01() - 0.5) * 4.0 * M_PI : k < 5 ? (u01() - 0.5) * 2000.0 : k == 5 ? (u01() - 0.5) * std::ldexp(1.0, -(int)(u01() * 60))
                 : (double)(int)(u01() * 64 - 32) * M_PI_2 + (u01() - 0.5) * 1e-6;
        if (!same(sca_gm::g_sin(a), p_sin(a))) bad[0]++;
        if (!same(sca_gm::g_cos(a), p_cos(a))) bad[1]++;
        // arctangents: both components over fourteen decades, every quadrant, quotients on both sides of 1/16 and of 1
        const double y = (u01() - 0.5) * std::pow(10.0, u01() * 14 - 9), x = (u01() - 0.5) * std::pow(10.0, u01() * 14 - 9);
        if (!same(sca_gm::g_atan2(y, x), p_atan2(y, x))) bad[2]++;
        const double cth = k < 6 ? u01() * 2.0 - 1.0 : (k == 6 ? 1.0 - u01() * 1e-9 : -1.0 + u01() * 1e-9);
        if (!same(sca_gm::g_acos(cth), p_acos(cth))) bad[3]++;
        const double b = (u01() - 0.5) * std::pow(10.0, u01() * 12 - 6);
        if (!same(sca_gm::g_pow2(b), p_pow(b, 2.0))) bad[4]++;
    }
    int64_t total = 0;
    for (int q = 0; q < 5; q++) { g_libm_bad[q] = bad[q]; total += bad[q]; }
    g_libm_state = total ? 1 : 0;
    if (g_libm_state && !getenv("SCA_QUIET"))
        fprintf(stderr, "[libsca_hip] note: this host's libm differs from the restated glibc 2.35 x86-64 FMA build in %lld of 20480 test "
                        "arguments (sin %lld, cos %lld, atan2 %lld, acos %lld, pow(x,2) %lld): the Dubins tracker reproduces the bits of THAT "
                        "libm (the one the golden vectors were recorded with); a Python reference run on this host would differ from both "
                        "in the last bit of some path lengths.\n",
                (long long)total, (long long)bad[0], (long long)bad[1], (long long)bad[2], (long long)bad[3], (long long)bad[4]);
}
int sca_libm_check(int64_t *mismatches /*5, nullable*/) {
    const int r = libm_check_run();
    if (mismatches) for (int q = 0; q < 5; q++) mismatches[q] = g_libm_bad[q];
    return r;
}

// ---- native v_pref tracker (host only, no GPU needed): scaPolicy.py:264-338 + dubinsmaneuver2d/3d.py ----------------
void *sca_tracker_create(int n, const double *goal, const double *goal_heading, const double *pref_speed,
                         const uint8_t *zaxis, double turning_radius, double pitch_min, double pitch_max,
                         double neighbor_dist) {
    if (n <= 0 || !goal || !goal_heading || !pref_speed) return nullptr;
    (void)libm_check_run();                                              // one note on stderr when this host's libm is another build
    auto *T = new sca_dubins::Tracker();
    T->n = n;
    T->goal.assign(goal, goal + 3 * (size_t)n);
    T->goal_heading.assign(goal_heading, goal_heading + 3 * (size_t)n);
    T->pref_speed.assign(pref_speed, pref_speed + n);
    T->zaxis.assign((size_t)n, 0);
    if (zaxis) T->zaxis.assign(zaxis, zaxis + n);
    T->turning_radius = turning_radius; T->pitchlims[0] = pitch_min; T->pitchlims[1] = pitch_max;
    T->neighbor_dist = neighbor_dist;
    T->st.assign((size_t)n, sca_dubins::AgentTrack());
    return T;
}
int sca_tracker_set_neighbor_dist(void *tr, const double *neighbor_dist /*n, nullable: back to the one value*/) {
    if (!tr) return SCA_ERR_ARG;
    auto *T = (sca_dubins::Tracker *)tr;
    if (neighbor_dist) T->nd_per_agent.assign(neighbor_dist, neighbor_dist + T->n); else T->nd_per_agent.clear();
    return 0;
}
int sca_tracker_set_agent_params(void *tr, const double *turning_radius, const double *pitch_lo, const double *pitch_hi) {
    if (!tr) return SCA_ERR_ARG;
    auto *T = (sca_dubins::Tracker *)tr;
    auto put = [&](std::vector<double> &v, const double *a) { if (a) v.assign(a, a + T->n); else v.clear(); };
    put(T->R_pa, turning_radius); put(T->plo_pa, pitch_lo); put(T->phi_pa, pitch_hi);
    return 0;
}
void sca_tracker_destroy(void *tr) {
    auto *T = (sca_dubins::Tracker *)tr;
    if (T) { delete T->pool; delete T; }
}
int sca_tracker_vpref(void *tr, const double *pos, const float *vel, const double *heading, const uint8_t *active,
                      const double *nbr0_dsq, double *vpref_out, int nthreads) {
    if (!tr || !pos || !vel || !heading || !active || !nbr0_dsq || !vpref_out) return SCA_ERR_ARG;
    sca_dubins::step_all(*(sca_dubins::Tracker *)tr, pos, vel, heading, active, nbr0_dsq, vpref_out, nthreads);
    return 0;
}
// diagnostics: the tracker record of one agent as 24 doubles (host tracker / device tracker): horizontal maneuver r_min, t, p,
// length; vertical r_min, t, p, length; plan length, sampling size, 0, 0, next, count, now_goal[3], v_pref[3], words
static void track_dump(const sca_dubins::AgentTrack &a, double *o) {
    const sca_dubins::Plan3D &P = a.plan;
    o[0] = P.h.r_min; o[1] = P.h.t; o[2] = P.h.p; o[3] = P.h.length; o[4] = P.v.r_min; o[5] = P.v.t; o[6] = P.v.p; o[7] = P.v.length;
    o[8] = P.length; o[9] = P.sampling_size; o[10] = (double)P.rounds; o[11] = 0.0; o[12] = (double)a.next; o[13] = (double)P.count;
    for (int q = 0; q < 3; q++) { o[14 + q] = a.now_goal[q]; o[17 + q] = a.v_pref[q]; }
    o[20] = P.mode[0] * 65536.0 + P.mode[1] * 256.0 + P.mode[2]; o[21] = P.mode[3] * 65536.0 + P.mode[4] * 256.0 + P.mode[5];
    o[22] = 64.0 * P.iters; o[23] = (double)a.replans;
}
int sca_tracker_debug(void *tr, int agent, double *out24) {
    if (!tr || !out24) return SCA_ERR_ARG;
    auto *T = (sca_dubins::Tracker *)tr;
    if (agent < 0 || agent >= T->n) return SCA_ERR_ARG;
    track_dump(T->st[agent], out24);
    return 0;
}
int sca_tracker_replans(void *tr, int32_t *replans) {
    if (!tr || !replans) return SCA_ERR_ARG;
    auto *T = (sca_dubins::Tracker *)tr;
    for (int i = 0; i < T->n; i++) replans[i] = T->st[i].replans;
    return 0;
}
int sca_dubins_plan(const double *qi5, const double *qf5, double rmin, double pitch_min, double pitch_max, double *length,
                    char *mode7, int32_t *n_samples, double *samples, int cap) {
    if (!qi5 || !qf5 || !length || !mode7) return SCA_ERR_ARG;
    const double pl[2] = {pitch_min, pitch_max};
    const sca_dubins::Plan3D P = sca_dubins::plan3d(qi5, qf5, rmin, pl);
    if (!P.ok) return SCA_ERR_STATE;
    *length = P.length;
    std::memcpy(mode7, P.mode, 7);
    const int cnt = (int)P.count;
    if (n_samples) *n_samples = cnt;
    if (samples) for (int i = 0; i < std::min(cnt, cap); i++) P.sample(i, samples + 5 * (size_t)i);
    return 0;
}

// host self-test: the lean search of the device's lane-per-plan kernels (sca_dubins.hpp, plan3d_lean, compiled for the host)
// against the literal plan3d on the given poses, bit for bit
int sca_selftest_plan3d_lean(int n, const double *q, double turning_radius, double pitch_lo, double pitch_hi, int64_t *mismatches,
                             int64_t *lean_candidates, int64_t *literal_candidates) {
    if (n < 0 || !q || !mismatches) return SCA_ERR_ARG;
    const double pl[2] = {pitch_lo, pitch_hi};
    auto same = [](double a, double b) { return std::memcmp(&a, &b, sizeof(double)) == 0 || (a != a && b != b); };
    int64_t bad = 0;
    sca_dubins::lean::g_host_fast = sca_dubins::lean::g_host_literal = 0;
    for (int i = 0; i < n; i++) {
        const double *qi = q + 10 * (size_t)i, *qf = qi + 5;
        const sca_dubins::Plan3D A = sca_dubins::plan3d(qi, qf, turning_radius, pl), B = sca_dubins::plan3d_lean(qi, qf, turning_radius, pl);
        bool ok = A.ok == B.ok && A.iters == B.iters && A.count == B.count && same(A.length, B.length) && same(A.sampling_size, B.sampling_size) &&
                  std::memcmp(A.mode, B.mode, 7) == 0;
        const sca_dubins::Maneuver2D *ma[2] = {&A.h, &A.v}, *mb[2] = {&B.h, &B.v};
        for (int k = 0; k < 2 && ok && A.ok; k++)
            ok = same(ma[k]->r_min, mb[k]->r_min) && same(ma[k]->t, mb[k]->t) && same(ma[k]->p, mb[k]->p) && same(ma[k]->length, mb[k]->length) &&
                 same(ma[k]->yaw, mb[k]->yaw);
        if (!ok) bad++;
    }
    *mismatches = bad;
    if (lean_candidates) *lean_candidates = sca_dubins::lean::g_host_fast;
    if (literal_candidates) *literal_candidates = sca_dubins::lean::g_host_literal;
    return 0;
}

// host self-test: the sign-parametrised CSC word of the device's four-lane planner against the literal word(), bit for bit
int sca_selftest_dubins_words(int n, const double *alpha, const double *beta, const double *d, int64_t *mismatches) {
    if (n < 0 || !alpha || !beta || !d || !mismatches) return SCA_ERR_ARG;
    int64_t bad = 0;
    auto same = [](double a, double b) { return std::memcmp(&a, &b, sizeof(double)) == 0 || (a != a && b != b); };
    for (int i = 0; i < n; i++) {
        double sa, sb, ca, cb;
        sca_dubins::m_sincos(alpha[i], sa, ca);
        sca_dubins::m_sincos(beta[i], sb, cb);
        const double c_ab = sca_dubins::m_cos(alpha[i] - beta[i]);
        const double mb = sca_dubins::mod2pi(beta[i]);
        for (int w = 0; w < 4; w++) {
            double t0 = 0, p0 = 0, q0 = 0, t1 = 0, p1 = 0, q1 = 0;
            char mode[3];
            const bool ok0 = sca_dubins::word(w, alpha[i], beta[i], d[i], sa, sb, ca, cb, c_ab, t0, p0, q0, mode);
            const bool ok1 = sca_dubins::csc_word_uniform(w, alpha[i], beta[i], mb, d[i], sa, sb, ca, cb, c_ab, t1, p1, q1);
            if (ok0 != ok1 || (ok0 && !(same(t0, t1) && same(p0, p1) && same(q0, q1)))) bad++;
        }
    }
    *mismatches = bad;
    return 0;
}

// A threshold measured on the 256-CU part (1024 SIMDs), scaled to this device: they are all "so many wavefronts per SIMD".
static inline int per_simd(const sca_ctx *c, long long at_1024_simds) { return (int)std::min<long long>(INT_MAX, at_1024_simds * c->simds / 1024); }
// ---- the same tracker on the device (sca_tracker.hip.h) ---------------------------------------------------------------
static int launch_tracker(sca_ctx *c, bool from_lists, bool in_pass);
static int part_free(sca_ctx *c);
static int part_classify(sca_ctx *c);
static int agent_params_clear(sca_ctx *c);
static int tracker_free(sca_ctx *c) {
    if (!c->trk.st) { c->trk_on = false; return 0; }
    CHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->trk.st); (void)hipFree(c->trk.nbr0); (void)hipFree(c->trk.list); (void)hipFree(c->trk.count); (void)hipFree(c->trk.bcount);
    (void)hipFree(c->trk_goal_heading);
    if (c->trk_R_pa) { (void)hipFree(c->trk_R_pa); c->trk_R_pa = nullptr; }
    if (c->trk_cls) { (void)hipFree(c->trk_cls); c->trk_cls = nullptr; }
    if (c->trk_plo_pa) { (void)hipFree(c->trk_plo_pa); c->trk_plo_pa = nullptr; }
    if (c->trk_phi_pa) { (void)hipFree(c->trk_phi_pa); c->trk_phi_pa = nullptr; }
    c->trk_many = false;
    c->trk_classes.clear();
    if (c->trk_stream) { (void)hipStreamSynchronize(c->trk_stream); (void)hipStreamDestroy(c->trk_stream); c->trk_stream = nullptr; }
    if (c->trk_fork) { (void)hipEventDestroy(c->trk_fork); c->trk_fork = nullptr; }
    if (c->trk_join) { (void)hipEventDestroy(c->trk_join); c->trk_join = nullptr; }
    if (c->trk_count_ev) { (void)hipEventDestroy(c->trk_count_ev); c->trk_count_ev = nullptr; }
    if (c->trk_host_count) { (void)hipHostFree(c->trk_host_count); c->trk_host_count = nullptr; }
    c->trk_count_pending = false; c->trk_last_count = -1;
    c->kd.skip_prep = 0;
    c->trk = TrackDev{}; c->trk_goal_heading = nullptr; c->trk_on = false; c->trk_in_pass = false;
    c->d.trk_nbr0 = nullptr;
    // the tracked agents go back to their policy's own v_pref rule (rvo3dPolicy.py:182-196): until round 5 they kept vpref_mode = 1 and the
    // tracker's LAST output, frozen -- bench.py's `solver_only` legs ran on that instead of the straight-line rule they are labelled with
    if (c->d.vpref_mode && c->n > 0) { CHK(c, hipMemsetAsync(c->d.vpref_mode, 0, c->n, c->stream)); CHK(c, hipStreamSynchronize(c->stream)); }
    return 0;
}
int sca_device_tracker_enable(sca_ctx *c, const double *goal_heading, double turning_radius, double pitch_min, double pitch_max,
                              int in_pass) {
    API_ENTER(c);
    if (!c->agents_set) { c->err = "sca_set_agents first"; return SCA_ERR_STATE; }
    ARG(c, goal_heading && turning_radius > 0);
    if (int r = tracker_free(c)) return r;
    const int n = c->n;
    CHK(c, hipMalloc((void **)&c->trk.st, sizeof(sca_dubins::AgentTrack) * n));
    CHK(c, hipMalloc((void **)&c->trk.nbr0, sizeof(double) * n));
    CHK(c, hipMalloc((void **)&c->trk.list, sizeof(int32_t) * (size_t)n * TRK_BUCKETS));
    CHK(c, hipMalloc((void **)&c->trk.count, sizeof(int32_t) * 4));
    CHK(c, hipMalloc((void **)&c->trk.bcount, sizeof(int32_t) * 4 * TRK_BUCKETS));
    c->trk.n = n;
    CHK(c, hipMalloc((void **)&c->trk_goal_heading, sizeof(double) * 3 * n));
    CHK(c, hipStreamCreateWithFlags(&c->trk_stream, hipStreamNonBlocking));
    CHK(c, hipEventCreateWithFlags(&c->trk_fork, hipEventDisableTiming));
    CHK(c, hipEventCreateWithFlags(&c->trk_join, hipEventDisableTiming));
    CHK(c, hipEventCreateWithFlags(&c->trk_count_ev, hipEventDisableTiming));
    CHK(c, hipHostMalloc((void **)&c->trk_host_count, sizeof(int) * 2));
    c->trk_count_pending = false; c->trk_last_count = -1;
    std::vector<sca_dubins::AgentTrack> init((size_t)n);
    std::vector<double> nb((size_t)n, -1.0);
    std::vector<uint8_t> pol((size_t)n), mode((size_t)n);
    CHK(c, hipMemcpyAsync(pol.data(), c->d.policy, n, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipMemcpyAsync(c->trk.st, init.data(), sizeof(sca_dubins::AgentTrack) * n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->trk.nbr0, nb.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->trk_goal_heading, goal_heading, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemsetAsync(c->trk.count, 0, sizeof(int32_t) * 4, c->stream));
    CHK(c, hipMemsetAsync(c->trk.bcount, 0, sizeof(int32_t) * 4 * TRK_BUCKETS, c->stream));
    CHK(c, hipMemsetAsync(c->d.nbr_valid, 0, n, c->stream));            // no policy pass of this agent set has left lists yet
    CHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) mode[i] = (pol[i] == SCA_POLICY_SCA || pol[i] == SCA_POLICY_RVO3D_DUBINS) ? 1 : 0;
    CHK(c, hipMemcpyAsync(c->d.vpref_mode, mode.data(), n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemsetAsync(c->d.vpref_ext, 0, sizeof(double) * 3 * n, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    c->trk.parity = 0;
    c->trk_view = sca_dubins::TrackView{c->d.goal, c->trk_goal_heading, c->d.pref_speed, c->d.zaxis, turning_radius, pitch_min,
                                        pitch_max, c->P.neighbor_dist, c->ap_nd, nullptr, nullptr, nullptr, nullptr, 0};
    c->trk_classes.clear();
    c->trk_many = false;
    c->trk_enable_vals[0] = turning_radius; c->trk_enable_vals[1] = pitch_min; c->trk_enable_vals[2] = pitch_max;
    {   // resolve the tracker's kernels now: the first launch of a kernel pays for looking it up in the code object, and the
        // forms are picked while the episode runs (k_track_replan's first launch used to fall into a timed step)
        hipFuncAttributes fa;
        for (const void *f : {(const void *)k_track, (const void *)k_replan, (const void *)k_replan_group<64>, (const void *)k_replan_group<32>,
                              (const void *)k_replan_group<16>, (const void *)k_replan_group<4>,
                              (const void *)k_track_replan, (const void *)k_solve_sweep, (const void *)k_solve_pick4})
            (void)hipFuncGetAttributes(&fa, f);
    }
    (void)libm_check_run();              // informational: one note on stderr, the verdict through sca_libm_check(); sca_last_error stays for failures
    c->trk_on = true; c->trk_in_pass = in_pass != 0;
    c->d.trk_nbr0 = c->trk_in_pass ? c->trk.nbr0 : nullptr;
    c->trk_passes = 0;
    c->trk_serial = getenv("SCA_TRACKER_SERIAL") != nullptr;
    c->trk_quad = getenv("SCA_TRACKER_NOQUAD") == nullptr;
    c->trk_group_fuse = getenv("SCA_TRACKER_NOGROUPFUSE") == nullptr;   // k_track_group for shards of <= TRK_SPEC4_MAX agents
    c->trk_fuse = getenv("SCA_TRACKER_FUSE") != nullptr;          // k_track_replan (no list, hence no ordering by expected length): opt-in since round 3
    c->trk.mid_max = getenv("SCA_TRK_MID_MAX") ? atoi(getenv("SCA_TRK_MID_MAX")) : per_simd(c, TRK_MID_MAX);
    c->trk.spec2_max = getenv("SCA_TRK_SPEC2_MAX") ? atoi(getenv("SCA_TRK_SPEC2_MAX")) : per_simd(c, TRK_SPEC2_MAX);
    c->trk.spec3_max = getenv("SCA_TRK_SPEC3_MAX") ? atoi(getenv("SCA_TRK_SPEC3_MAX")) : per_simd(c, TRK_SPEC3_MAX);
    c->trk.spec4_max = getenv("SCA_TRK_SPEC4_MAX") ? atoi(getenv("SCA_TRK_SPEC4_MAX")) : per_simd(c, TRK_SPEC4_MAX);
    return 0;
}
// agent.turning_radius / agent.pitchlims per agent (scaPolicy.py:95,272,302 read the agent's own).  Arrays of n, any of them NULL = the value of
// sca_device_tracker_enable for everybody; all NULL: back to that one value.  Untracked agents' entries (policy not SCA / RVO3D+Dubins) are ignored.
// Up to TRK_MAX_CLASSES distinct (R, lo, hi) among the tracked agents: CLASSES -- the re-plan kernels run once per class with the class's
// values as kernel arguments (scalar registers throughout the search).  More than that (the reference has no limit, agent.py:24-29): the
// PER-AGENT form -- every re-plan gets a wavefront of its own (k_replan_group<64> / k_track_group at any count), which loads ITS agent's
// three values into scalar registers; slower than the lane-per-plan forms for large counts, never refused.
constexpr size_t TRK_MAX_CLASSES = 16;
int sca_device_tracker_set_agent_params(sca_ctx *c, int n, const double *turning_radius, const double *pitch_lo, const double *pitch_hi) {
    API_ENTER(c);
    if (!c->trk_on) { c->err = "sca_device_tracker_enable first"; return SCA_ERR_STATE; }
    CHK(c, hipStreamSynchronize(c->stream));
    // (every call starts from the enable-time values: a one-class call overwrites the view's scalars, and "any of them NULL" / "all NULL"
    // promise sca_device_tracker_enable's values, not the previous call's -- ADVICE r5)
    c->trk_view.turning_radius = c->trk_enable_vals[0]; c->trk_view.pitch_lo = c->trk_enable_vals[1]; c->trk_view.pitch_hi = c->trk_enable_vals[2];
    c->trk_classes.clear();
    c->trk_many = false;
    c->trk_view.R_pa = nullptr; c->trk_view.plo_pa = nullptr; c->trk_view.phi_pa = nullptr; c->trk_view.cls = nullptr; c->trk_view.class_id = 0;
    if (!turning_radius && !pitch_lo && !pitch_hi) return 0;
    ARG(c, n == c->n);
    // tracked = by POLICY, as the kernels decide it (tracker_owns), not by the v_pref mode of the moment (sca_set_vpref may change that later)
    std::vector<uint8_t> pol((size_t)n), cls((size_t)n, 0);
    CHK(c, hipMemcpy(pol.data(), c->d.policy, (size_t)n, hipMemcpyDeviceToHost));
    std::vector<std::array<double, 3>> classes;
    std::vector<double> R((size_t)n), LO((size_t)n), HI((size_t)n);
    bool many = false;
    for (int i = 0; i < n; i++) {
        const std::array<double, 3> v = {turning_radius ? turning_radius[i] : c->trk_enable_vals[0], pitch_lo ? pitch_lo[i] : c->trk_enable_vals[1],
                                         pitch_hi ? pitch_hi[i] : c->trk_enable_vals[2]};
        R[i] = v[0]; LO[i] = v[1]; HI[i] = v[2];
        if (pol[i] != SCA_POLICY_SCA && pol[i] != SCA_POLICY_RVO3D_DUBINS) continue;     // not a tracked agent
        if (!(std::isfinite(v[0]) && v[0] > 0.0) || !std::isfinite(v[1]) || !std::isfinite(v[2]) || !(v[1] < v[2])) {
            c->err = "sca_device_tracker_set_agent_params: agent " + std::to_string(i) + " has a turning radius / pitch limits out of range (R > 0, pitch_lo < pitch_hi)";
            return SCA_ERR_ARG;
        }
        if (many) continue;
        size_t k = 0;
        while (k < classes.size() && classes[k] != v) k++;
        if (k == classes.size()) {
            if (classes.size() == TRK_MAX_CLASSES) { many = true; continue; }
            classes.push_back(v);
        }
        cls[i] = (uint8_t)k;
    }
    if (!c->trk_R_pa) { CHK(c, hipMalloc((void **)&c->trk_R_pa, sizeof(double) * (size_t)c->max_n)); CHK(c, hipMalloc((void **)&c->trk_cls, (size_t)c->max_n)); }
    CHK(c, hipMemcpy(c->trk_R_pa, R.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    c->trk_view.R_pa = c->trk_R_pa;
    if (many) {
        if (!c->trk_plo_pa) {
            CHK(c, hipMalloc((void **)&c->trk_plo_pa, sizeof(double) * (size_t)c->max_n));
            CHK(c, hipMalloc((void **)&c->trk_phi_pa, sizeof(double) * (size_t)c->max_n));
        }
        CHK(c, hipMemcpy(c->trk_plo_pa, LO.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
        CHK(c, hipMemcpy(c->trk_phi_pa, HI.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
        c->trk_view.plo_pa = c->trk_plo_pa; c->trk_view.phi_pa = c->trk_phi_pa;
        c->trk_many = true;
        return 0;
    }
    CHK(c, hipMemcpy(c->trk_cls, cls.data(), (size_t)n, hipMemcpyHostToDevice));
    if (classes.size() > 1) { c->trk_classes = classes; c->trk_view.cls = c->trk_cls; }
    else if (!classes.empty()) {                                        // one class after all: its values in the scalars, no filter
        c->trk_view.turning_radius = classes[0][0]; c->trk_view.pitch_lo = classes[0][1]; c->trk_view.pitch_hi = classes[0][2];
    }
    return 0;
}
int sca_device_tracker_disable(sca_ctx *c) {
    API_ENTER(c);
    return tracker_free(c);
}
int sca_device_tracker_vpref(sca_ctx *c, const double *nbr0_dsq, double *vpref_out) {
    API_ENTER(c);
    if (!c->trk_on || !c->state_set) { c->err = "sca_device_tracker_enable and sca_set_state first"; return SCA_ERR_STATE; }
    if (nbr0_dsq) CHK(c, hipMemcpyAsync(c->trk.nbr0, nbr0_dsq, sizeof(double) * c->n, hipMemcpyHostToDevice, c->stream));
    if (int r = launch_tracker(c, nbr0_dsq == nullptr, false)) return r;
    if (vpref_out) CHK(c, hipMemcpyAsync(vpref_out, c->d.vpref_ext, sizeof(double) * 3 * c->n, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int sca_device_tracker_debug(sca_ctx *c, int agent, double *out24) {
    if (!c || !out24) return SCA_ERR_ARG;
    if (int r_ = api_enter(c)) return r_;
    if (!c->trk_on || agent < 0 || agent >= c->n) { c->err = "no device tracker / bad agent"; return SCA_ERR_STATE; }
    sca_dubins::AgentTrack a;
    CHK(c, hipStreamSynchronize(c->stream));
    CHK(c, hipMemcpy(&a, c->trk.st + agent, sizeof(a), hipMemcpyDeviceToHost));
    track_dump(a, out24);
    return 0;
}
int sca_device_tracker_replans(sca_ctx *c, int32_t *replans) {
    API_ENTER(c);
    ARG(c, replans);
    if (!c->trk_on) { c->err = "sca_device_tracker_enable first"; return SCA_ERR_STATE; }
    int32_t *tmp = (int32_t *)c->trk.list;                              // free between passes
    hipLaunchKernelGGL(k_track_replans, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->trk.st, tmp, c->n);
    CHK(c, hipMemcpyAsync(replans, tmp, sizeof(int32_t) * c->n, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int sca_create(const sca_params *p, int device, int max_agents, int max_obstacles, sca_ctx **out) {
    if (!out || max_agents <= 0 || max_obstacles < 0) return SCA_ERR_ARG;
    sca_ctx *c = new sca_ctx();
    *out = c;
    sca_params def, own;
    sca_default_params_v2(&def, (int32_t)sizeof(sca_params));
    if (!p) p = &def;
    if (p->struct_bytes < (int32_t)(offsetof(sca_params, dt_nominal) + sizeof(double))) {     // a version-100 struct: 56 bytes, no dt_nominal
        memcpy(&own, p, offsetof(sca_params, dt_nominal));
        own.struct_bytes = (int32_t)sizeof(sca_params);
        own.dt_nominal = own.time_step;
        p = &own;
    }
    if (p->max_neighbors < 1 || p->max_neighbors > SCA_MAX_NEIGHBORS) { c->err = "max_neighbors out of range (1 .. 16)"; return SCA_ERR_ARG; }
    {   // the kernels divide by these, size grid cells with them and bisect acos on max_heading_change: refuse what they were not built for
        auto pos = [](double x) { return std::isfinite(x) && x > 0.0; };
        const char *bad = !pos(p->neighbor_dist) ? "neighbor_dist" : !pos(p->time_step) ? "time_step" : !pos(p->time_horizon) ? "time_horizon"
                        : !pos(p->max_speed) ? "max_speed" : !pos(p->dt_nominal) ? "dt_nominal"
                        : !(std::isfinite(p->near_goal_threshold) && p->near_goal_threshold >= 0.0) ? "near_goal_threshold"
                        : !(p->max_heading_change >= 0.0 && p->max_heading_change <= M_PI) ? "max_heading_change" : nullptr;
        if (bad) { c->err = std::string("sca_params.") + bad + " out of range"; return SCA_ERR_ARG; }
    }
    c->device = device; c->max_n = max_agents; c->max_m = max_obstacles;
    c->P.neighbor_dist = p->neighbor_dist; c->P.time_step = p->time_step; c->P.time_horizon = p->time_horizon;
    c->P.max_speed = p->max_speed; c->P.max_heading_change = p->max_heading_change;
    c->P.near_goal_threshold = p->near_goal_threshold; c->P.max_neighbors = p->max_neighbors; c->P.pad = 0; c->P.dt_nominal = p->dt_nominal;
    c->P.range_sq = sca_gm::g_pow2(c->P.neighbor_dist);                 // neighborDist ** 2 (scaPolicy.py:112): glibc's pow, restated
    c->P.cos_heading_thr = cos_threshold(p->max_heading_change);
    c->P_ctx = c->P;
    if (const char *e = std::getenv("SCA_K1_PACKED")) c->k1_force = std::atoi(e) != 0;    // A/B switch for measurements
    if (const char *e = std::getenv("SCA_AUTO_BACKOFF_DIV")) c->auto_div = std::min(64, std::max(1, std::atoi(e)));
    c->auto_no_tail = std::getenv("SCA_AUTO_NO_TAIL") != nullptr;
    if (const char *e = std::getenv("SCA_AUTO_TAIL_MAX")) c->auto_tail_max = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("SCA_SOLVE_SPLIT")) c->solve_split = std::atoi(e) != 0;
    if (const char *e = std::getenv("SCA_KD_NOHINT")) c->kd_nohint = std::atoi(e) != 0;
    if (const char *e = std::getenv("SCA_KD_TAIL_LEVEL")) c->kd_tail_level = std::atoi(e);
    if (const char *e = std::getenv("SCA_KD_TOP")) c->kd_top = std::atoi(e) != 0;
    if (const char *e = std::getenv("SCA_EXT_STOP")) c->ext_stop = std::atoi(e) != 0;
    if (const char *e = std::getenv("SCA_KD_TICKET")) c->kd_force_ticket = std::atoi(e) != 0;
    if (const char *e = std::getenv("SCA_KD_WAVE_CAP")) c->kd_wave_cap = std::min(KD_WAVE_CAP, std::max(2 * KD_WAVE_FLOOR, std::atoi(e)));
    int ndev = 0;
    CHK(c, hipGetDeviceCount(&ndev));
    if (ndev <= 0) { c->err = "no HIP device: libsca_hip has no CPU path"; return SCA_ERR_HIP; }
    CHK(c, hipSetDevice(device));
    CHK(c, hipStreamCreateWithFlags(&c->stream_own, hipStreamNonBlocking));
    c->stream = c->stream_own;
    {
        hipDeviceProp_t prop;
        CHK(c, hipGetDeviceProperties(&prop, device));
        c->cus = std::max(1, prop.multiProcessorCount);
        c->simds = 4 * c->cus;
        c->solve_fb_max = std::getenv("SCA_SOLVE_FB_MAX") ? std::atoi(std::getenv("SCA_SOLVE_FB_MAX")) : per_simd(c, 2048);
        c->action_fb_max = std::getenv("SCA_ACTION_FB_MAX") ? std::atoi(std::getenv("SCA_ACTION_FB_MAX")) : per_simd(c, 16384);
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_kd_lv_rank<false>, KD_LV_T, 0) == hipSuccess && per_cu > 0)
            c->kd_rank_capacity = per_cu * c->cus;
        else (void)hipGetLastError();
    }
    for (auto &e : c->ev) CHK(c, hipEventCreate(&e));
    CHK(c, hipEventCreateWithFlags(&c->kd_ev, hipEventDisableTiming));
    CHK(c, hipHostMalloc((void **)&c->kd_host_counts, sizeof(int) * (2 * KD_MAX_LEVELS + 3)));   // counts | nchunks
    const size_t N = (size_t)max_agents, M = (size_t)max_obstacles;
    DeviceView &d = c->d;
    int r = 0;
    r |= dalloc(c, &c->rec_own, N); d.rec = c->rec_own;
    r |= dalloc(c, &c->rec_new_own, N); d.rec_new = c->rec_new_own;
    r |= dalloc(c, &d.heading, 3 * N); r |= dalloc(c, &d.goal, 3 * N); r |= dalloc(c, &d.pref_speed, N);
    r |= dalloc(c, &d.vpref_ext, 3 * N); r |= dalloc(c, &d.total_dist, N); r |= dalloc(c, &d.max_run_dist, N);
    r |= dalloc(c, &d.step_num, N); r |= dalloc(c, &d.vpref_mode, N); r |= dalloc(c, &d.policy, N);
    r |= dalloc(c, &d.zaxis, N); r |= dalloc(c, &c->lp_list, N);
    r |= dalloc(c, &d.obs, M); r |= dalloc(c, &d.atree, 2 * N); r |= dalloc(c, &d.aperm, N);
    r |= dalloc(c, &d.obs_sorted, M); r |= dalloc(c, &d.awide, 2 * N); r |= dalloc(c, &d.owide, 2 * M);
    r |= dalloc(c, &d.otree, 2 * M); r |= dalloc(c, &d.operm, M);
    r |= dalloc(c, &d.nbr_n, N); r |= dalloc(c, &d.nbr_id, N * K_MAX); r |= dalloc(c, &d.nbr_dsq, N * K_MAX);
    r |= dalloc(c, &d.coll_new, N); r |= dalloc(c, &d.nbr_valid, N);
    r |= dalloc(c, &d.near_n, N); r |= dalloc(c, &d.near_id, N * NEAR_MAX);
    r |= dalloc(c, &d.action, N * 8); r |= dalloc(c, &d.vpref_used, 3 * N); r |= dalloc(c, &d.vpost, 3 * N);
    r |= dalloc(c, &d.fb_list, N); r |= dalloc(c, &d.fb_count, 1); r |= dalloc(c, &d.is_fb, N);
    { Prep *pp = nullptr; r |= dalloc(c, &pp, N); d.prep = pp; } r |= dalloc(c, &d.diag, N * 8);
    r |= dalloc(c, &d.status, N); r |= dalloc(c, &d.done_count, 256 * 32); r |= dalloc(c, &d.agent_steps, 256 * 16);
    r |= dalloc(c, &c->kd.kx, N); r |= dalloc(c, &c->kd.ky, N); r |= dalloc(c, &c->kd.kz, N);
    r |= dalloc(c, &c->kd.mr, N);
    d.kx = c->kd.kx; d.ky = c->kd.ky; d.kz = c->kd.kz;
    c->kd.job_cap = (int)(N / 64 + 64);
    r |= dalloc(c, &c->kd.jobs[0], (size_t)c->kd.job_cap); r |= dalloc(c, &c->kd.jobs[1], (size_t)c->kd.job_cap);
    r |= dalloc(c, &c->kd.small, N); r |= dalloc(c, &c->kd.counts, (size_t)2 * KD_MAX_LEVELS + 4); r |= dalloc(c, &c->kd.ticket, (size_t)KD_MAX_LEVELS + 1);   // counts | nchunks (one readback) | tail slot counter
    c->kd.chunk_cap = (int)(N / KD_CHUNK + N / KD_WAVE_FLOOR + 8);
    r |= dalloc(c, &c->kd.nbox, (size_t)2 * c->kd.job_cap * 6); r |= dalloc(c, &c->kd.nge, (size_t)2 * c->kd.job_cap);
    r |= dalloc(c, &c->kd.ps, N);
    r |= dalloc(c, &c->kd.cbox, (size_t)2 * c->kd.job_cap * 12); r |= dalloc(c, &c->kd.chain, (size_t)c->kd.chunk_cap);
    r |= dalloc(c, &c->kd.chunks[0], (size_t)c->kd.chunk_cap); r |= dalloc(c, &c->kd.chunks[1], (size_t)c->kd.chunk_cap);
    c->kd.nchunks = c->kd.counts ? c->kd.counts + KD_MAX_LEVELS + 2 : nullptr;
    {   // grid of SCA_NBR_GRID: at least two buckets per agent
        GridDev &g = c->grid;
        g.hbits = 10;
        while (((size_t)1 << g.hbits) < 2 * N) g.hbits++;
        const size_t H = (size_t)1 << g.hbits;
        r |= dalloc(c, &g.count, H); r |= dalloc(c, &g.range, H); r |= dalloc(c, &g.cursor, 1);
        r |= dalloc(c, &g.bucket, N); r |= dalloc(c, &g.slot, N);
        r |= dalloc(c, &g.gx, N); r |= dalloc(c, &g.gy, N); r |= dalloc(c, &g.gz, N);
        r |= dalloc(c, &g.gid, N); r |= dalloc(c, &g.gkey, N);
        g.skip_prep = 0;
        g.inv_cell = grid_inv_cell(c->P.neighbor_dist);
    }
    if (!r) {   // the root's accumulators start empty (every build's last kernel resets them for the next one)
        unsigned long long h[12];
        const double pinf = INFINITY, ninf = -INFINITY;
        unsigned long long kp, kn;
        { unsigned long long u; std::memcpy(&u, &pinf, 8); kp = (u >> 63) ? ~u : (u | 0x8000000000000000ull); }
        { unsigned long long u; std::memcpy(&u, &ninf, 8); kn = (u >> 63) ? ~u : (u | 0x8000000000000000ull); }
        for (int q = 0; q < 12; q++) h[q] = (q % 6) < 3 ? kp : kn;
        CHK(c, hipMemcpyAsync(c->kd.nbox, h, sizeof(unsigned long long) * 6, hipMemcpyHostToDevice, c->stream));
        CHK(c, hipMemcpyAsync(c->kd.cbox, h, sizeof(unsigned long long) * 12, hipMemcpyHostToDevice, c->stream));
        CHK(c, hipStreamSynchronize(c->stream));
    }
    if (r) return SCA_ERR_HIP;
    // candidate tables: [unit256 (768) | unit128 (384) | phi256 (256) | phi128 (128)]
    std::vector<double> tab(768 + 384 + 256 + 128);
    candidate_table_host(256, tab.data(), tab.data() + 768 + 384);
    candidate_table_host(128, tab.data() + 768, tab.data() + 768 + 384 + 256);
    // k_solve's posture pre-filter treats the table directions as unit vectors (|u|^2 = 1 inside its 1e-13 margin): make sure
    for (int set = 0; set < 2; set++) {
        const int nn = set ? 128 : 256;
        const double *u = tab.data() + (set ? 768 : 0);
        for (int i = 0; i < nn; i++) {
            const double n2 = u[i] * u[i] + u[nn + i] * u[nn + i] + u[2 * nn + i] * u[2 * nn + i];
            if (std::fabs(n2 - 1.0) > 1e-14) { c->err = "candidate table is not unit length"; return SCA_ERR_STATE; }
        }
    }
    if (dalloc(c, &c->tab, tab.size())) return SCA_ERR_HIP;
    CHK(c, hipMemcpyAsync(c->tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    d.unit256 = c->tab; d.unit128 = c->tab + 768; d.phi256 = c->tab + 768 + 384; d.phi128 = c->tab + 768 + 384 + 256;
    d.n = 0; d.m = 0; d.shard_begin = 0; d.shard_count = 0;
    return 0;
}

void sca_destroy(sca_ctx *c) {
    if (!c) return;
    DeviceView &d = c->d;
    if (c->stream_own) (void)hipStreamSynchronize(c->stream_own);
    if (c->comm) { (void)hipStreamSynchronize(c->stream); (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    (void)tracker_free(c);
    (void)part_free(c);
    if (c->h_done) { (void)hipHostFree(c->h_done); c->h_done = nullptr; }
    if (c->ap_dev) { (void)hipFree(c->ap_dev); (void)hipFree(c->ap_nd); c->ap_dev = nullptr; c->ap_nd = nullptr; }
    void *ptrs[] = {c->rec_own, c->rec_new_own, d.heading, d.goal, d.pref_speed, d.vpref_ext, d.total_dist, d.max_run_dist,
                    d.step_num, d.vpref_mode, d.policy, d.zaxis, d.obs, d.obs_sorted, d.awide, d.owide, d.atree, d.aperm, d.otree, d.operm, d.nbr_n,
                    d.nbr_id, d.nbr_dsq, d.coll_new, d.nbr_valid, d.near_n, d.near_id, d.action, d.vpref_used, d.vpost, d.fb_list, d.fb_count, d.is_fb, d.prep, d.diag, d.status,
                    d.done_count, d.agent_steps, c->tab, c->kd.kx, c->kd.ky, c->kd.kz, c->kd.mr,
                    c->kd.jobs[0], c->kd.jobs[1], c->kd.small, c->kd.counts, c->kd.nbox, c->kd.nge, c->kd.ps, c->kd.cbox, c->kd.chain, c->kd.chunks[0], c->kd.chunks[1], d.hist,
                    c->grid.count, c->grid.range, c->grid.cursor, c->grid.bucket, c->grid.slot, c->grid.gx, c->grid.gy, c->grid.gz,
                    c->grid.gid, c->grid.gkey, c->lp_list, d.sw_slot, d.sw_surv, d.sw_n};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->pool) (void)hipEventDestroy(e);
    for (auto &e : c->pool_trk) (void)hipEventDestroy(e);
    for (auto &e : c->pool_xch) (void)hipEventDestroy(e);
    for (auto &e : c->pool_kd) (void)hipEventDestroy(e);
    if (c->kd_stream) { (void)hipStreamSynchronize(c->kd_stream); (void)hipStreamDestroy(c->kd_stream); }
    for (hipEvent_t e : {c->ev_auto_fork, c->ev_auto_k1g, c->ev_auto_moved, c->ev_auto_cnt}) if (e) (void)hipEventDestroy(e);   // (ev_auto_kd aliases ev_auto_kdq[])
    if (c->kdq_list) (void)hipFree(c->kdq_list);
    if (c->kdq_count) (void)hipFree(c->kdq_count);
    if (c->kdq_host) (void)hipHostFree(c->kdq_host);
    if (c->auto_busy) (void)hipFree(c->auto_busy);
    if (c->auto_sync) (void)hipFree(c->auto_sync);
    if (c->auto_ticket) (void)hipFree(c->auto_ticket);
    if (c->d.kdq_stats) (void)hipFree(c->d.kdq_stats);
    for (hipEvent_t e : c->ev_auto_gather) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_auto_kdq) if (e) (void)hipEventDestroy(e);
    if (c->kd_ev) (void)hipEventDestroy(c->kd_ev);
    if (c->kd_host_counts) (void)hipHostFree(c->kd_host_counts);
    if (c->stream_own) (void)hipStreamDestroy(c->stream_own);
    delete c;
}

int sca_set_obstacles(sca_ctx *c, int m, const double *pos, const double *radius) {
    API_ENTER(c);
    ARG(c, m >= 0 && m <= c->max_m);
    ARG(c, m == 0 || (pos && radius));
    c->m = m; c->d.m = m;
    c->max_obs_radius = 0;
    for (int i = 0; i < m; i++) c->max_obs_radius = std::max(c->max_obs_radius, radius[i]);
    if (m == 0) return 0;
    std::vector<ObsRec> h(m);
    for (int i = 0; i < m; i++) { h[i].px = pos[3 * i]; h[i].py = pos[3 * i + 1]; h[i].pz = pos[3 * i + 2]; h[i].radius = radius[i]; }
    std::vector<int32_t> perm(m);
    for (int i = 0; i < m; i++) perm[i] = i;                           // kdTree.py:51-52
    std::vector<KdNode> tree;
    kd_build_host(m, pos, perm.data(), tree);                          // mampenv.py:20, built once
    CHK(c, hipMemcpyAsync(c->d.obs, h.data(), sizeof(ObsRec) * m, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d.operm, perm.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d.otree, tree.data(), sizeof(KdNode) * (2 * m - 1), hipMemcpyHostToDevice, c->stream));
    std::vector<KdWide> wide;
    kd_widen_host(m, tree, wide);
    std::vector<ObsRec> sorted(m);
    for (int i = 0; i < m; i++) sorted[i] = h[perm[i]];
    CHK(c, hipMemcpyAsync(c->d.owide, wide.data(), sizeof(KdWide) * (2 * m - 1), hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d.obs_sorted, sorted.data(), sizeof(ObsRec) * m, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

static int agent_params_clear(sca_ctx *c) {
    if (!c->ap_dev) return 0;
    CHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->ap_dev); (void)hipFree(c->ap_nd);
    c->ap_dev = nullptr; c->ap_nd = nullptr; c->d.ap = nullptr;
    c->P = c->P_ctx;
    c->grid.inv_cell = grid_inv_cell(c->P.neighbor_dist);
    if (c->trk_on) c->trk_view.nd_per_agent = nullptr;
    return 0;
}
// The reference keeps maxNeighbors / neighborDist / timeStep / timeHorizon / maxSpeed / max_heading_change / dt_nominal on every Agent object
// (agent.py:24-41) and every policy reads its own agent's (scaPolicy.py:112, util.py:8,17, orca3dPolicyOfficial.py:44,98,108, agent.py:87-99,
// mampenv.py:90-92).  Arrays of n = the agents of sca_set_agents; a NULL array keeps the context's value for everybody; n = 0 (or all NULL)
// returns to one value per context.  The grid's cells are sized for the largest neighborDist, the collision reach for the largest step.
int sca_set_agent_params(sca_ctx *c, int n, const double *neighbor_dist, const int32_t *max_neighbors, const double *time_step,
                         const double *time_horizon, const double *max_speed, const double *max_heading_change, const double *dt_nominal) {
    API_ENTER(c);
    if (!c->agents_set) { c->err = "sca_set_agents first"; return SCA_ERR_STATE; }
    const bool any = neighbor_dist || max_neighbors || time_step || time_horizon || max_speed || max_heading_change || dt_nominal;
    // (with the cell-owner partition on: only while the grid's cell size stays what the slabs were cut with -- checked below)
    if (n == 0 || !any) {
        if (c->part_on && c->ap_dev && grid_inv_cell(c->P_ctx.neighbor_dist) != c->part.inv_cell) {
            c->err = "sca_set_agent_params: this would change the grid's cell size under the cell-owner partition's slabs: sca_partition_disable first"; return SCA_ERR_STATE;
        }
        return agent_params_clear(c);
    }
    ARG(c, n == c->n);
    std::vector<AgentPar> h((size_t)n);
    std::vector<double> nd((size_t)n);
    auto pos = [](double x) { return std::isfinite(x) && x > 0.0; };
    std::vector<std::pair<double, double>> thr_cache;               // (max_heading_change, its cosine threshold): few distinct values
    Params env = c->P_ctx;
    double nd_max = 0, ms_max = 0, dt_max = 0;
    for (int i = 0; i < n; i++) {
        AgentPar a;
        a.neighbor_dist = neighbor_dist ? neighbor_dist[i] : c->P_ctx.neighbor_dist;
        a.time_step = time_step ? time_step[i] : c->P_ctx.time_step;
        a.time_horizon = time_horizon ? time_horizon[i] : c->P_ctx.time_horizon;
        a.max_speed = max_speed ? max_speed[i] : c->P_ctx.max_speed;
        a.dt_nominal = dt_nominal ? dt_nominal[i] : c->P_ctx.dt_nominal;
        a.max_neighbors = max_neighbors ? max_neighbors[i] : c->P_ctx.max_neighbors;
        a.pad = 0;
        a.range_sq = sca_gm::g_pow2(a.neighbor_dist);
        const double mhc = max_heading_change ? max_heading_change[i] : c->P_ctx.max_heading_change;
        if (!pos(a.neighbor_dist) || !pos(a.time_step) || !pos(a.time_horizon) || !pos(a.max_speed) || !pos(a.dt_nominal) ||
            a.max_neighbors < 1 || a.max_neighbors > SCA_MAX_NEIGHBORS || !(mhc >= 0.0 && mhc <= M_PI)) {
            c->err = "sca_set_agent_params: agent " + std::to_string(i) + " has an attribute out of range (see sca_params)";
            return SCA_ERR_ARG;
        }
        a.cos_heading_thr = c->P_ctx.cos_heading_thr;
        if (max_heading_change) {
            bool hit = false;
            for (auto &e : thr_cache) if (e.first == mhc) { a.cos_heading_thr = e.second; hit = true; break; }
            if (!hit) { a.cos_heading_thr = cos_threshold(mhc); thr_cache.push_back({mhc, a.cos_heading_thr}); }
        }
        h[i] = a; nd[i] = a.neighbor_dist;
        nd_max = std::max(nd_max, a.neighbor_dist); ms_max = std::max(ms_max, a.max_speed); dt_max = std::max(dt_max, a.dt_nominal);
    }
    if (c->part_on && grid_inv_cell(nd_max) != c->part.inv_cell) {
        c->err = "sca_set_agent_params: the largest neighbor_dist would change the grid's cell size under the cell-owner partition's slabs: call it "
                 "before sca_partition_init (or sca_partition_disable first)";
        return SCA_ERR_STATE;
    }
    CHK(c, hipStreamSynchronize(c->stream));
    if (!c->ap_dev) { CHK(c, hipMalloc((void **)&c->ap_dev, sizeof(AgentPar) * (size_t)c->max_n)); CHK(c, hipMalloc((void **)&c->ap_nd, sizeof(double) * (size_t)c->max_n)); }
    CHK(c, hipMemcpy(c->ap_dev, h.data(), sizeof(AgentPar) * (size_t)n, hipMemcpyHostToDevice));
    CHK(c, hipMemcpy(c->ap_nd, nd.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    c->d.ap = c->ap_dev;
    // the context's own Params become the envelope: what sizes the grid's cells (largest range) and the collision reach (largest step)
    env.neighbor_dist = nd_max; env.max_speed = ms_max; env.dt_nominal = dt_max; env.range_sq = sca_gm::g_pow2(nd_max);
    c->P = env;
    c->grid.inv_cell = grid_inv_cell(c->P.neighbor_dist);
    if (c->trk_on) c->trk_view.nd_per_agent = c->ap_nd;
    c->near_valid = false;
    return 0;
}

int sca_set_agents(sca_ctx *c, int n, const double *radius, const double *pref_speed, const double *goal,
                   const uint8_t *policy, const uint8_t *zaxis, const double *max_run_dist) {
    API_ENTER(c);
    ARG(c, n > 0 && n <= c->max_n);
    ARG(c, radius && pref_speed && goal && policy && max_run_dist);
    for (int i = 0; i < n; i++) ARG(c, policy[i] <= SCA_POLICY_RVO3D_DUBINS);        // (before anything of the previous set is torn down)
    if (c->d.hist) {                                                  // the log's pitch is n: a new agent set starts a new log
        CHK(c, hipStreamSynchronize(c->stream));
        CHK(c, hipFree(c->d.hist));
        c->d.hist = nullptr; c->d.hist_cap = 0; c->d.hist_row = 0;
    }
    if (int r = tracker_free(c)) return r;                            // the tracker records belong to the old agent set
    if (int r = agent_params_clear(c)) return r;                      // ... and so do the agents' own solver attributes
    if (int r = part_free(c)) return r;                               // ... and so do the partition's lists
    if (c->comm && n % c->comm_nranks) { c->err = "agent count must be a multiple of the communicator's rank count"; return SCA_ERR_ARG; }
    c->n = n; c->d.n = n; c->d.shard_begin = 0; c->d.shard_count = n;
    if (c->comm) { c->d.shard_count = n / c->comm_nranks; c->d.shard_begin = c->comm_rank * c->d.shard_count; }
    c->h_rec.assign(n, PubRec{});
    c->max_radius = 0; c->max_pref_speed = 0;
    for (int i = 0; i < n; i++) {
        c->h_rec[i].radius = radius[i];
        c->max_radius = std::max(c->max_radius, radius[i]);
        c->max_pref_speed = std::max(c->max_pref_speed, pref_speed[i]);
    }
    c->h_lp_list.clear();
    for (int i = 0; i < n; i++) if (policy[i] == SCA_POLICY_ORCA3D_LP) c->h_lp_list.push_back(i);
    c->d.lp_kernel = 0;                                               // decided per pass from the shard's LP agent count
    if (!c->h_lp_list.empty())
        CHK(c, hipMemcpyAsync(c->lp_list, c->h_lp_list.data(), sizeof(int32_t) * c->h_lp_list.size(), hipMemcpyHostToDevice, c->stream));
    c->h_perm.resize(n);
    for (int i = 0; i < n; i++) c->h_perm[i] = i;                     // kdTree.py:43-45
    c->perm_on_device = false;
    std::vector<uint8_t> z(n, 0), mode(n, 0);
    if (zaxis) z.assign(zaxis, zaxis + n);
    CHK(c, hipMemcpyAsync(c->d.pref_speed, pref_speed, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d.goal, goal, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d.policy, policy, n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d.zaxis, z.data(), n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d.vpref_mode, mode.data(), n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d.max_run_dist, max_run_dist, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemsetAsync(c->d.total_dist, 0, sizeof(double) * n, c->stream));
    CHK(c, hipMemsetAsync(c->d.step_num, 0, sizeof(int32_t) * n, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    c->agents_set = true; c->state_set = false; c->h_pos_valid = false;
    c->kdq_last = -1; c->auto_backoff = 0; c->kd_ahead = false; c->auto_ran = false;
    return 0;
}

int sca_set_state(sca_ctx *c, const double *pos, const float *vel, const double *heading, const uint8_t *flags,
                  const double *total_dist, const int32_t *step_num) {
    API_ENTER(c);
    if (!c->agents_set) { c->err = "sca_set_agents first"; return SCA_ERR_STATE; }
    ARG(c, pos && vel && heading && flags);
    const int n = c->n;
    for (int i = 0; i < n; i++) {
        PubRec &r = c->h_rec[i];
        r.px = pos[3 * i]; r.py = pos[3 * i + 1]; r.pz = pos[3 * i + 2];
        r.vx = vel[3 * i]; r.vy = vel[3 * i + 1]; r.vz = vel[3 * i + 2];
        r.flags = flags[i];
    }
    c->h_pos.assign(pos, pos + 3 * (size_t)n);
    c->kd_single_hint = 0; c->kd_gen++;   // a new state: the previous trees' depth profile says nothing
    c->kd_ahead = false; c->kdq_last = -1; c->auto_backoff = 0;           // ... and neither do the AUTO passes' counts (no tree is built ahead
                                                                          // across calls: sca_run_steps consumes its own or abandons it)
    c->h_pos_valid = true;
    c->near_valid = false;
    CHK(c, hipMemcpyAsync(c->d.rec, c->h_rec.data(), sizeof(PubRec) * n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d.heading, heading, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
    if (total_dist) CHK(c, hipMemcpyAsync(c->d.total_dist, total_dist, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    if (step_num) CHK(c, hipMemcpyAsync(c->d.step_num, step_num, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    c->state_set = true; c->state_fresh = true;
    if (c->part_on) return part_classify(c);                              // a complete state again: ownership follows from it
    return 0;
}

static int fetch_records(sca_ctx *c) {
    CHK(c, hipMemcpyAsync(c->h_rec.data(), c->d.rec, sizeof(PubRec) * c->n, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int sca_get_state(sca_ctx *c, double *pos, float *vel, double *heading, uint8_t *flags, double *total_dist,
                  int32_t *step_num) {
    API_ENTER(c);
    if (!c->state_set) { c->err = "no state"; return SCA_ERR_STATE; }
    const int n = c->n;
    if (int r = fetch_records(c)) return r;
    for (int i = 0; i < n; i++) {
        const PubRec &r = c->h_rec[i];
        if (pos) { pos[3 * i] = r.px; pos[3 * i + 1] = r.py; pos[3 * i + 2] = r.pz; }
        if (vel) { vel[3 * i] = r.vx; vel[3 * i + 1] = r.vy; vel[3 * i + 2] = r.vz; }
        if (flags) flags[i] = (uint8_t)r.flags;
    }
    if (heading) CHK(c, hipMemcpyAsync(heading, c->d.heading, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, c->stream));
    if (total_dist) CHK(c, hipMemcpyAsync(total_dist, c->d.total_dist, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    if (step_num) CHK(c, hipMemcpyAsync(step_num, c->d.step_num, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int sca_set_kd_perm(sca_ctx *c, const int32_t *perm) {
    API_ENTER(c);
    ARG(c, perm && c->agents_set);
    c->h_perm.assign(perm, perm + c->n);
    c->perm_on_device = false;
    return 0;
}
int sca_get_kd_perm(sca_ctx *c, int32_t *perm) {
    API_ENTER(c);
    ARG(c, perm && c->agents_set);
    if (c->perm_on_device) {
        CHK(c, hipMemcpyAsync(c->h_perm.data(), c->d.aperm, sizeof(int32_t) * c->n, hipMemcpyDeviceToHost, c->stream));
        CHK(c, hipStreamSynchronize(c->stream));
    }
    std::memcpy(perm, c->h_perm.data(), sizeof(int32_t) * c->n);
    return 0;
}

int sca_get_kd_tree(sca_ctx *c, double *tree_out) {
    API_ENTER(c);
    ARG(c, tree_out && c->agents_set);
    const int n = c->n;
    std::vector<KdNode> t((size_t)2 * n, KdNode{});
    if (!c->perm_on_device) {
        // host-built tree: the node array itself was uploaded
        CHK(c, hipMemcpyAsync(t.data(), c->d.atree, sizeof(KdNode) * (2 * n - 1), hipMemcpyDeviceToHost, c->stream));
        CHK(c, hipStreamSynchronize(c->stream));
    } else {
        // device-built tree: the build only writes the query records (header + both children's boxes) and the root's
        // node; the node array of kdTree.py:14-21 is put together here
        std::vector<KdWide> w((size_t)2 * n);
        CHK(c, hipMemcpyAsync(w.data(), c->d.awide, sizeof(KdWide) * (2 * n - 1), hipMemcpyDeviceToHost, c->stream));
        CHK(c, hipMemcpyAsync(t.data(), c->d.atree, sizeof(KdNode), hipMemcpyDeviceToHost, c->stream));
        CHK(c, hipStreamSynchronize(c->stream));
        std::vector<int> st{0};
        while (!st.empty()) {
            const int i = st.back();
            st.pop_back();
            t[i].begin = w[i].begin; t[i].end = w[i].end; t[i].left = w[i].left; t[i].right = w[i].right;
            if (w[i].end - w[i].begin > MAX_LEAF) {
                const int ch[2] = {w[i].left, w[i].right};
                for (int sd = 0; sd < 2; sd++) {
                    if (ch[sd] <= i || ch[sd] >= 2 * n - 1) { c->err = "corrupt kd tree"; return SCA_ERR_STATE; }
                    for (int k = 0; k < 3; k++) { t[ch[sd]].mn[k] = w[i].bx[kdw_idx(sd, 0, k)]; t[ch[sd]].mx[k] = w[i].bx[kdw_idx(sd, 1, k)]; }
                    st.push_back(ch[sd]);
                }
            } else { t[i].left = 0; t[i].right = 0; }
        }
    }
    for (int i = 0; i < 2 * n - 1; i++) {
        double *o = tree_out + 10 * (size_t)i;
        o[0] = t[i].begin; o[1] = t[i].end; o[2] = t[i].left; o[3] = t[i].right;
        for (int k = 0; k < 3; k++) { o[4 + k] = t[i].mn[k]; o[7 + k] = t[i].mx[k]; }
    }
    return 0;
}

int sca_set_vpref(sca_ctx *c, const double *vpref, const uint8_t *mode) {
    API_ENTER(c);
    ARG(c, vpref && mode && c->agents_set);
    CHK(c, hipMemcpyAsync(c->d.vpref_ext, vpref, sizeof(double) * 3 * c->n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d.vpref_mode, mode, c->n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// KDTree.buildAgentTree (mampenv.py:28) on the HOST from a position read-back (SCA_NBR_KDTREE_HOSTBUILD: the A/B reference of
// the device build below, which is what SCA_NBR_KDTREE runs).
static int build_agent_tree(sca_ctx *c) {
    const int n = c->n;
    if (!c->h_pos_valid) {
        if (int r = fetch_records(c)) return r;
        c->h_pos.resize(3 * (size_t)n);
        for (int i = 0; i < n; i++) { c->h_pos[3 * i] = c->h_rec[i].px; c->h_pos[3 * i + 1] = c->h_rec[i].py; c->h_pos[3 * i + 2] = c->h_rec[i].pz; }
        c->h_pos_valid = true;
    }
    kd_build_host(n, c->h_pos.data(), c->h_perm.data(), c->h_tree);
    CHK(c, hipMemcpyAsync(c->d.atree, c->h_tree.data(), sizeof(KdNode) * (2 * n - 1), hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->d.aperm, c->h_perm.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
    std::vector<KdWide> wide;
    kd_widen_host(n, c->h_tree, wide);
    std::vector<double> k((size_t)3 * n);
    for (int p = 0; p < n; p++)
        for (int a = 0; a < 3; a++) k[(size_t)a * n + p] = c->h_pos[3 * (size_t)c->h_perm[p] + a];
    CHK(c, hipMemcpyAsync(c->d.awide, wide.data(), sizeof(KdWide) * (2 * n - 1), hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->kd.kx, k.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->kd.ky, k.data() + n, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(c->kd.kz, k.data() + 2 * (size_t)n, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));                             // the staging vectors die with this scope
    CHK(c, hipMemsetAsync(c->d.done_count, 0, sizeof(int32_t) * 256 * 32, c->stream));
    return 0;
}

// KDTree.buildAgentTree on the device (sca_kdbuild.hip.h): gather, one launch per level of large nodes, one launch
// that finishes every small subtree.  Everything is enqueued on the context's stream; nothing comes back to the host.
static int build_agent_tree_device(sca_ctx *c, hipStream_t ks, const DeviceView &d) {
    const int n = c->n;
    // (no size limit: a level with more chunks than the chip holds at once takes its chunks by arrival, k_kd_lv_rank<true>; rounds
    // 1-2 refused more than 256 CUs x 4 workgroups x KD_CHUNK = 2 097 152 agents, with the CU count as a literal)
    if (!c->perm_on_device) {
        CHK(c, hipMemcpyAsync(d.aperm, c->h_perm.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, ks));
        c->perm_on_device = true;
    }
    // size of the subtrees handed to k_kd_block: 1.25 x the average node size of the first level that fits (n / 2^k), so that
    // the nodes of that level -- all within a few per cent of the average -- are on one side of it
    // Subtrees of up to 1536 members (one workgroup of 12 wavefronts each) when the build has the chip to itself; up to 1024 when it
    // runs beside the tracker's re-plans (a side stream): twice as many, smaller workgroups spread over twice as many CUs, each
    // competing with fewer re-plan wavefronts -- measured at c5 (N = 16 384, 16 subtrees of ~1024 against 32 of ~512): step 0.284
    // -> 0.268 ms; c3 (no tracker, 4 against 8 subtrees) the other way round: 0.127 against 0.129
    // a tree of up to KT_M members: its top by one workgroup in LDS (k_kd_top), which is cheap enough per level to go one level
    // further down than the level passes would -- subtrees of ~512 instead of ~1024 members for k_kd_block
    // (not beside the tracker's re-plans, whose ~250-register wavefronts sit on every SIMD: a workgroup of sixteen wavefronts and 139 KB
    // of LDS then waits for room and for issue slots -- measured as a middle tier under the level passes: 60 us instead of 30 at c5)
    const bool beside = c->trk_on && c->trk_in_pass && ks != c->stream;
    const bool top = c->kd_top && n <= KT_M && c->kd_tail_level < 0 && !beside;
    const int cap = c->kd_wave_cap > 0 ? c->kd_wave_cap : (top ? 768 : (beside ? 1024 : KD_WAVE_CAP));
    int wave_max = (n <= 1024 && cap >= 1024) ? 1024 : cap; // a tree that fits one workgroup: the smaller one if it can
    if (n > cap) {
        double sz = (double)n;
        while (sz > cap / 1.25) sz *= 0.5;
        wave_max = std::min(cap, std::max(cap / 2 + 1, (int)std::ceil(1.25 * sz)));
    }
    c->kd.wave_max = wave_max;
    hipEvent_t kb0 = nullptr, kb1 = nullptr;                            // profiling: the build's device time on the stream it runs on, every 16th build
    if (c->profiling && (c->kd_builds & 15u) == 0 && c->pool_kd_used + 2 <= 2 * 1024) {
        for (hipEvent_t *ev : {&kb0, &kb1}) {
            if (c->pool_kd_used == (int)c->pool_kd.size()) { hipEvent_t n_; CHK(c, hipEventCreate(&n_)); c->pool_kd.push_back(n_); }
            *ev = c->pool_kd[c->pool_kd_used++];
        }
        CHK(c, hipEventRecord(kb0, ks));
    }
    if (top && KT_M / (wave_max + 1) >= KT_NODES) { c->err = "k_kd_top: wave_max below its table bound"; return SCA_ERR_STATE; }   // (static_assert'ed unreachable)
    if (c->kd.aux) {                                                    // SCA_NBR_AUTO: the last kernel of the build that reads the record buffer
        LAUNCH_REC(c, c->ev_auto_gather[c->auto_builds & 1u], k_kd_gather, dim3((n + 255) / 256), dim3(256), ks, d, c->kd, c->P);
        c->auto_builds++;
    } else hipLaunchKernelGGL(k_kd_gather, dim3((n + 255) / 256), dim3(256), 0, ks, d, c->kd, c->P);
    int levels = 0;
    if (n > wave_max && top) {
        hipLaunchKernelGGL(k_kd_top, dim3(1), dim3(KT_T), 0, ks, d, c->kd);
        levels = 1;
    } else if (n > wave_max) {
        // Level passes: two launches per level (rank | swap) while the nodes span several chunks, then ONE launch
        // (k_kd_level_tail) in which every remaining node's workgroup finishes its whole subtree down to wave_max.  Where the
        // switch happens only sets the speed -- the tail handles any node size and any depth -- so it is taken from the
        // statistics of an earlier build when they have arrived (the first level whose nodes all fit one chunk), otherwise
        // from the balanced tree.
        int first_single = 1;
        while (((long long)KD_CHUNK << first_single) < n) first_single++;       // n / 2^l <= KD_CHUNK
        first_single += 1;                                                      // uneven midpoint splits
        if (c->kd_ev_pending && hipEventQuery(c->kd_ev) == hipSuccess && c->kd_ev_gen != c->kd_gen) c->kd_ev_pending = false;   // stale
        if (c->kd_ev_pending && hipEventQuery(c->kd_ev) == hipSuccess) {
            int depth = 0;
            while (depth < KD_MAX_LEVELS && c->kd_host_counts[depth] > 0) depth++;
            const int *nch = c->kd_host_counts + KD_MAX_LEVELS + 2;
            int single = depth;
            while (single > 0 && nch[single - 1] == c->kd_host_counts[single - 1]) single--;
            c->kd_single_hint = c->kd_host_counts[KD_MAX_LEVELS + 1] ? 0 : single + 1;        // 1-based
            c->kd_ev_pending = false;
        }
        if (c->kd_single_hint > 0 && !c->kd_nohint) first_single = c->kd_single_hint - 1;
        if (c->kd_tail_level >= 0) first_single = c->kd_tail_level;
        first_single = std::min(first_single, KD_MAX_LEVELS - 2);
        levels = first_single + 1;
        const int grid = std::min(c->kd.chunk_cap, n / KD_CHUNK + n / (wave_max / 2 + 1) + 8);   // >= chunks of any level of n agents (a node of the level passes has > wave_max members... its children > 0)
        for (int l = 0; l < first_single; l++) {
            // chunk by arrival once a level can have more chunks than are resident at once (k_kd_lv_rank); SCA_KD_TICKET=1 forces it (tests)
            if (grid > c->kd_rank_capacity || c->kd_force_ticket)
                hipLaunchKernelGGL(k_kd_lv_rank<true>, dim3(grid), dim3(KD_LV_T), 0, ks, c->kd, l, ++c->kd_token);
            else
                hipLaunchKernelGGL(k_kd_lv_rank<false>, dim3(grid), dim3(KD_LV_T), 0, ks, c->kd, l, ++c->kd_token);
            hipLaunchKernelGGL(k_kd_lv_swap, dim3(grid), dim3(KD_LV_T + 64), 0, ks, d, c->kd, l);   // + the bookkeeping wavefront
        }
        hipLaunchKernelGGL(k_kd_level_tail, dim3(grid), dim3(KD_LV_T), 0, ks, d, c->kd, first_single, ++c->kd_token);
    }

    const int sgrid = std::max(1, std::min(1024, 4 * n / wave_max + 2));
    // one workgroup per subtree, in LDS: the smallest form that holds wave_max members (two positions per thread)
    // (LAUNCH_OPT: an AUTO build in the tail form records the pass's kd-query event behind this kernel)
    if (wave_max <= 256) LAUNCH_OPT(c, c->kd_block_stop, (k_kd_block<256, 128>), dim3(sgrid), dim3(128), ks, d, c->kd, levels, c->kd_tail_arg);
    else if (wave_max <= 512) LAUNCH_OPT(c, c->kd_block_stop, (k_kd_block<512, 256>), dim3(sgrid), dim3(256), ks, d, c->kd, levels, c->kd_tail_arg);
    else if (wave_max <= 768) LAUNCH_OPT(c, c->kd_block_stop, (k_kd_block<768, 384>), dim3(sgrid), dim3(384), ks, d, c->kd, levels, c->kd_tail_arg);
    else if (wave_max <= 1024) LAUNCH_OPT(c, c->kd_block_stop, (k_kd_block<1024, 512>), dim3(sgrid), dim3(512), ks, d, c->kd, levels, c->kd_tail_arg);
    else if (wave_max <= 1280) LAUNCH_OPT(c, c->kd_block_stop, (k_kd_block<1280, 640>), dim3(sgrid), dim3(640), ks, d, c->kd, levels, c->kd_tail_arg);
    else LAUNCH_OPT(c, c->kd_block_stop, (k_kd_block<KD_WAVE_CAP, KD_WAVE_CAP / 2>), dim3(sgrid), dim3(KD_WAVE_CAP / 2), ks, d, c->kd, levels, c->kd_tail_arg);
    CHK(c, hipGetLastError());
    if (kb1) CHK(c, hipEventRecord(kb1, ks));
    // the tree's depth profile changes slowly: one small readback (a copy sits in the stream between the build and K1) every
    // 8th build, every build while no hint exists yet
    c->kd_builds++;
    if (n > wave_max && !top && !c->kd_ev_pending && (c->kd_single_hint == 0 || (c->kd_builds & 7u) == 0)) {
        CHK(c, hipMemcpyAsync(c->kd_host_counts, c->kd.counts, sizeof(int) * (2 * KD_MAX_LEVELS + 3), hipMemcpyDeviceToHost, ks));
        CHK(c, hipEventRecord(c->kd_ev, ks));
        c->kd_ev_pending = true;
        c->kd_ev_gen = c->kd_gen;
    }
    return 0;
}
// SCA_NBR_GRID: counting sort of all agents into cells of neighborDist (sca_grid.hip.h), three launches
static int build_agent_grid_device(sca_ctx *c) {
    const int n = std::max(256, c->part_on ? c->d.n_present : c->n);      // k_grid_count's first 256 lanes also reset the step's counters
    const int H = 1 << c->grid.hbits;
    hipLaunchKernelGGL(k_grid_count, dim3((n + 255) / 256), dim3(256), 0, c->nbr_stream, c->d, c->grid, c->P);
    hipLaunchKernelGGL(k_grid_alloc, dim3((H + 256 * GRID_ALLOC_PER - 1) / (256 * GRID_ALLOC_PER)), dim3(256), 0, c->nbr_stream, c->grid);
    hipLaunchKernelGGL(k_grid_fill, dim3((n + 255) / 256), dim3(256), 0, c->nbr_stream, c->d, c->grid);
    CHK(c, hipGetLastError());
    return 0;
}
static int check_kd_overflow(sca_ctx *c) {
    int flag = 0;
    CHK(c, hipMemcpyAsync(&flag, c->kd.counts + KD_MAX_LEVELS + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    if (flag) {
        c->kd_single_hint = 0;
        CHK(c, hipMemsetAsync(c->kd.counts + KD_MAX_LEVELS + 1, 0, sizeof(int), c->stream));      // reported: start afresh
        c->err = "device kd-tree build overflow (a build since the last check ran out of levels or table space): results are invalid; "
                 "use SCA_NBR_KDTREE_HOSTBUILD for this scene [code " + std::to_string(flag) + "]";
        return SCA_ERR_STATE;
    }
    return 0;
}

// A pair that touches after the move was within r_a + r_b + 2 * max_step of each other before it (an obstacle:
// r_a + r_o + max_step); l3norm rounds to 5 dp, hence the 1e-4.  The kernels add the agent's own radius.
static void collide_reach(const sca_ctx *c, double &agent_reach, double &obs_reach) {
    const double max_step = 1.01 * std::max(c->max_pref_speed, c->P.max_speed) * c->P.dt_nominal;
    agent_reach = c->max_radius + 2.0 * max_step + 1e-4;
    obs_reach = c->max_obs_radius + max_step + 1e-4;
}

static int pool_event(sca_ctx *c, hipEvent_t *out) {
    if (c->pool_used == (int)c->pool.size()) {
        hipEvent_t e;
        CHK(c, hipEventCreate(&e));
        c->pool.push_back(e);
    }
    *out = c->pool[c->pool_used++];
    return 0;
}

// v_pref of the SCA / RVO3D+Dubins agents of the shard, before anything of the pass reads it (the per-agent prologue does).
// in_pass = true (a policy pass with the tracker inside): agent.neighbors[0] comes from trk.nbr0, which the previous pass's
// epilogue saved -- so nothing here reads the neighbour lists, and the neighbour query of this pass may run beside it.
// in_pass = false (sca_device_tracker_vpref): from the lists as they are (from_lists) or from what the caller uploaded.
static int launch_tracker(sca_ctx *c, bool from_lists, bool in_pass) {
    const int cnt = c->d.shard_count;
    TrackDev K = c->trk;
    K.nbr0_from_lists = (from_lists && !in_pass) ? 1 : 0;
    if (!in_pass) K.prep = 0;
    // The device-side count of this pass decides which re-plan kernel does the work: every launched kernel reads it and returns
    // unless it falls into its range (lo, hi].  Launching all five every pass would cost four empty launches on the critical
    // path; the count of an earlier pass (copied back on the side stream, never waited for) says
    // which of them can be left out -- the ranges of those that are launched are widened so that every count is still somebody's
    // (a count that jumps is then re-planned by a form that is slower for it, never by nobody).  When nearly the whole shard
    // re-plans in the lane-per-plan form, k_track's list is not worth its launch either: k_track_replan does both.
    // (the first count of an episode is waited for once: a caller that enqueues a long burst of steps runs far ahead of the device,
    // and without it the whole burst would be launched with the forms of "count unknown")
    if (c->trk_count_pending && c->trk_last_count < 0) CHK(c, hipEventSynchronize(c->trk_count_ev));
    if (c->trk_count_pending && hipEventQuery(c->trk_count_ev) == hipSuccess) { c->trk_last_count = c->trk_host_count[0]; c->trk_count_pending = false; }
    // forms 0..3: k_replan_group<64 / 32 / 16 / 4> with the natural ranges (up[i - 1], up[i]]; form 4: one lane per plan, above
    const int lc = c->trk_last_count;
    const bool known = lc >= 0;
    int up[5] = {K.spec4_max, K.spec3_max, K.spec2_max, K.mid_max, INT_MAX};
    for (int i = 1; i < 4; i++) up[i] = std::max(up[i], up[i - 1]);
    bool want[5];
    int nwant = 0;
    for (int i = 0; i < 5; i++) {
        const long long lower = i ? up[i - 1] : -1, upper = up[i];
        const bool possible = cnt > lower && upper > lower && (i == 4 || c->trk_quad);
        want[i] = possible && (!known || (lc > lower - lower / 4 && (i == 4 || lc <= upper + upper / 4)));
        nwant += want[i];
    }
    if (!c->trk_quad) { for (int i = 0; i < 4; i++) want[i] = false; want[4] = true; nwant = 1; }
    if (nwant == 0) { want[4] = true; nwant = 1; }
    // the per-agent form (more classes of (turning radius, pitch limits) than launches are worth): a wavefront per plan at ANY count
    if (c->trk_many) { want[0] = true; for (int i = 1; i < 5; i++) want[i] = false; nwant = 1; }
    const bool lane = want[4];
    const bool fused = in_pass && lane && nwant == 1 && c->trk_fuse && !c->part_on && (long long)lc * 4 >= (long long)cnt * 3;
    // a shard of so few agents that each can have a wavefront (and a SIMD): decision and search in one launch (k_track_group)
    const bool group_fused = in_pass && (c->trk_quad || c->trk_many) && c->trk_group_fuse && !c->part_on && cnt <= K.spec4_max;
    if (group_fused) c->forms |= SCA_FORM_TRACK_FUSED | SCA_FORM_REPLAN_FEW;
    else c->forms |= (fused ? SCA_FORM_TRACK_FUSED : 0) | (nwant > (lane ? 1 : 0) ? SCA_FORM_REPLAN_FEW : 0) | (lane ? SCA_FORM_REPLAN_LANE : 0);
    if (!fused && !group_fused) hipLaunchKernelGGL(k_track, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, c->d, c->trk_view, K);
    hipStream_t rs = c->stream;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (c->profiling && (c->prof_tick & 15u) == 0 && c->pool_trk_used + 2 <= 2 * 4096) {
        for (hipEvent_t *e : {&t0, &t1}) {
            if (c->pool_trk_used == (int)c->pool_trk.size()) { hipEvent_t n_; CHK(c, hipEventCreate(&n_)); c->pool_trk.push_back(n_); }
            *e = c->pool_trk[c->pool_trk_used++];
        }
        CHK(c, hipEventRecord(t0, rs));
    }
    // tracked agents whose turning radius / pitch limits differ (sca_device_tracker_set_agent_params) are planned class by class: the same
    // launches once per class of equal values, each with ITS values in the view's scalars; an agent of another class leaves at once
    const int nclass = c->trk_classes.empty() ? 1 : (int)c->trk_classes.size();
    for (int cl = 0; cl < nclass; cl++) {
    sca_dubins::TrackView V = c->trk_view;
    if (!c->trk_classes.empty()) { V.class_id = cl; V.turning_radius = c->trk_classes[cl][0]; V.pitch_lo = c->trk_classes[cl][1]; V.pitch_hi = c->trk_classes[cl][2]; }
    if (group_fused) {
        K.lo = -1; K.hi = INT_MAX;
        hipLaunchKernelGGL(k_track_group, dim3((unsigned)(((long long)cnt * 64 + TRK_GROUP_THREADS - 1) / TRK_GROUP_THREADS)), dim3(TRK_GROUP_THREADS), 0,
                           rs, c->d, V, K);
    } else if (fused) {
        K.lo = -1; K.hi = INT_MAX;
        hipLaunchKernelGGL(k_track_replan, dim3((cnt + TRK_REPLAN_LANES - 1) / TRK_REPLAN_LANES), dim3(TRK_REPLAN_LANES), 0, rs, c->d,
                           V, K);
    } else {
        int prev_up = -1, left = nwant;
        for (int i = 0; i < 5; i++) {
            if (!want[i]) continue;
            left--;
            K.lo = prev_up; K.hi = left ? up[i] : INT_MAX;               // the first launched form starts at 0, the last one takes the rest
            prev_up = up[i];
            const int top = K.hi == INT_MAX ? cnt : std::min(cnt, up[i]);  // plans this launch must be able to hold
            if (top <= 0) continue;                                        // (a range moved to nothing by the tuning switches)
            // workgroups of four wavefronts, one per SIMD of a CU: the forms need <= 256 registers, so a SIMD can hold two of
            // their wavefronts, and with one-wavefront workgroups the dispatcher doubles some SIMDs up while others stay empty
            const dim3 blk(TRK_GROUP_THREADS);
            const auto blocks = [&](int lanes) { return dim3((unsigned)(((long long)top * lanes + TRK_GROUP_THREADS - 1) / TRK_GROUP_THREADS)); };
            switch (i) {
            case 0: hipLaunchKernelGGL(k_replan_group<64>, blocks(64), blk, 0, rs, c->d, V, K); break;
            case 1: hipLaunchKernelGGL(k_replan_group<32>, blocks(32), blk, 0, rs, c->d, V, K); break;
            case 2: hipLaunchKernelGGL(k_replan_group<16>, blocks(16), blk, 0, rs, c->d, V, K); break;
            case 3: hipLaunchKernelGGL(k_replan_group<4>, blocks(4), blk, 0, rs, c->d, V, K); break;
            default:
                hipLaunchKernelGGL(k_replan, dim3((cnt + TRK_REPLAN_LANES - 1) / TRK_REPLAN_LANES), dim3(TRK_REPLAN_LANES), 0, rs, c->d,
                                   V, K);
            }
        }
    }
    }
    if (t1) CHK(c, hipEventRecord(t1, rs));
    CHK(c, hipGetLastError());
    c->trk.parity = (c->trk.parity + 1) & 3;
    return 0;
}

// K3 form for this pass: one lane per agent (k_lp) once the shard has enough LP agents to fill the chip that way -- measured:
// 100 000 agents 65 vs 106 us, 4096 agents 23 vs 12 us (a lane alone needs ~12 us for its 16 planes and the LP) --, else the
// wave-per-agent form inside k_solve.  SCA_LP_FORM=lane|wave forces one (A/B measurements).
constexpr int LP_LANE_MIN = 16384;              // one lane per LP agent once they fill the chip: 16 agents per SIMD
static void choose_lp_form(sca_ctx *c, int &lo, int &hi) {
    if (c->part_on) {
        // ownership is dynamic: the LP kernels walk all owned agents and skip the others (the share of LP agents decides the form)
        lo = 0; hi = c->h_lp_list.empty() ? 0 : c->d.shard_count;
        c->d.lp_kernel = ((long long)c->h_lp_list.size() / std::max(1, c->part_nranks) >= per_simd(c, LP_LANE_MIN)) ? 1 : 0;
        return;
    }
    const auto b = std::lower_bound(c->h_lp_list.begin(), c->h_lp_list.end(), c->d.shard_begin);
    const auto e = std::lower_bound(c->h_lp_list.begin(), c->h_lp_list.end(), c->d.shard_begin + c->d.shard_count);
    lo = (int)(b - c->h_lp_list.begin()); hi = (int)(e - c->h_lp_list.begin());
    const char *f = getenv("SCA_LP_FORM");
    c->d.lp_kernel = (hi - lo >= per_simd(c, LP_LANE_MIN)) ? 1 : 0;
    if (f && f[0] == 'l') c->d.lp_kernel = hi > lo ? 1 : 0;
    if (f && f[0] == 'w') c->d.lp_kernel = 0;
}

// k_solve as k_solve_sweep (beside the re-plans) + k_solve_pick4 (behind them)?  It pays while the re-plans are the longer
// branch of the pass: the lane-per-plan kernel takes ~0.2 + 0.235 * (wavefronts per SIMD, rounded up) ms whatever the
// count inside a round, the neighbour chain grows with the shard.  Measured on the circle (96 % of the agents re-plan per
// step), shard sizes 24 576 ... 262 144: a gain of 5-10 % of the step up to 61 440 agents in the first round and up to
// ~114 000 in the second, a loss of 3-7 % elsewhere (the sweep then lengthens the branch that already ends last).
static bool choose_solve_split(const sca_ctx *c, bool overlap, int cnt) {
    if (c->solve_split >= 0) return c->solve_split != 0;
    if (!overlap) return false;
    const int est = c->trk_last_count >= 0 ? c->trk_last_count : cnt;    // re-plans of a recent pass (all agents before the first readback)
    if (est <= c->trk.mid_max) return false;      // the many-lanes-per-plan forms: short re-plans, nothing to hide behind (measured equal
                                                  // with and without at 18 000 .. 30 000 agents)
    const int per_round = 64 * c->simds;          // plans of the lane-per-plan kernel that are one wavefront per SIMD
    const int rounds = (est + per_round - 1) / per_round;
    return rounds == 1 ? cnt <= per_simd(c, 61440) : (rounds == 2 ? cnt <= per_simd(c, 114688) : false);
}

// SCA_NBR_AUTO's resources, on first use
// An SCA_NBR_AUTO step is bound by what the HOST needs to enqueue it (tools/gpu/exp_host.py: 79 us per step of 91 at N = 4096 -- twelve
// launches at ~2.3 us and ten event operations at ~5 us), so a cross-stream wait is only enqueued when the event it would wait for has
// not fired yet (in the steady state the kd stream is a whole pass ahead of what these guards protect; the query costs < 1 us).
static int wait_if_pending(sca_ctx *c, hipStream_t s, hipEvent_t e) {
    // (launch errors of the kernels enqueued before this call are sticky in hipGetLastError: look at them BEFORE the query below, whose
    // hipErrorNotReady has to be cleared -- ADVICE r5: a failed launch in front of a pending event used to be dropped with it)
    {
        const hipError_t le = hipGetLastError();                         // (an earlier query's hipErrorNotReady is an answer, not a failure)
        if (le != hipSuccess && le != hipErrorNotReady) CHK(c, le);
    }
    const hipError_t q = hipEventQuery(e);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) CHK(c, q);                                // a real error of the query is an error
    (void)hipGetLastError();                                             // (hipErrorNotReady is an answer, not a failure)
    CHK(c, hipStreamWaitEvent(s, e, 0));
    return 0;
}
static int auto_prepare(sca_ctx *c) {
    if (c->kd_stream) return 0;
    CHK(c, hipStreamCreateWithFlags(&c->kd_stream, hipStreamNonBlocking));
    for (hipEvent_t *e : {&c->ev_auto_fork, &c->ev_auto_k1g, &c->ev_auto_moved, &c->ev_auto_cnt})
        CHK(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    // two lists, alternating by pass: a pass's kd query may still be reading its list's length (to find it empty) when the next
    // pass's grid build resets and refills the other one
    CHK(c, hipMalloc((void **)&c->kdq_list, 2 * sizeof(int32_t) * (size_t)c->max_n));
    CHK(c, hipMalloc((void **)&c->kdq_count, 2 * sizeof(int32_t)));
    CHK(c, hipMemsetAsync(c->kdq_count, 0, 2 * sizeof(int32_t), c->stream));
    CHK(c, hipMalloc((void **)&c->auto_busy, sizeof(unsigned)));
    CHK(c, hipMemsetAsync(c->auto_busy, 0, sizeof(unsigned), c->stream));
    CHK(c, hipMalloc((void **)&c->d.kdq_stats, sizeof(unsigned long long) * 4));
    CHK(c, hipMemsetAsync(c->d.kdq_stats, 0, sizeof(unsigned long long) * 4, c->stream));
    CHK(c, hipMalloc((void **)&c->auto_ticket, sizeof(int)));
    CHK(c, hipMemsetAsync(c->auto_ticket, 0, sizeof(int), c->stream));
    CHK(c, hipMalloc((void **)&c->auto_sync, 8 * sizeof(unsigned)));
    CHK(c, hipMemsetAsync(c->auto_sync, 0, 8 * sizeof(unsigned), c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    for (hipEvent_t &e : c->ev_auto_gather) CHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t &e : c->ev_auto_kdq) CHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->auto_seq = 0; c->auto_builds = 0;
    c->auto_waitvalue = getenv("SCA_AUTO_EVENT_WAIT") == nullptr;
    c->auto_tail_ok = false;
    c->auto_tail_ok = c->auto_waitvalue && !c->auto_no_tail;
    c->kd_tail_seq = 0;

    CHK(c, hipHostMalloc((void **)&c->kdq_host, sizeof(int)));
    c->d.kdq_list = c->kdq_list; c->d.kdq_count = c->kdq_count; c->d.kdq_busy = c->auto_busy;
    c->kdq_pending = false; c->kdq_last = -1; c->auto_backoff = 0; c->kd_ahead = false;
    return 0;
}
// the kd build of a pass on kd_stream, beside whatever runs on the other streams (aux: it leaves the step's counters and the solve's
// prologue to the grid's kernels).  `positions`: the record buffer that holds the positions to build from.
static int auto_enqueue_kd_build(sca_ctx *c, hipEvent_t after, const PubRec *positions) {
    CHK(c, hipStreamWaitEvent(c->kd_stream, after, 0));
    DeviceView v = c->d;
    v.rec = const_cast<PubRec *>(positions);
#ifdef SCA_TIMELINE
    if (c->kd_ahead_enqueue) v.tl_step = c->d.tl_step + 1;              // a build enqueued ahead belongs to the NEXT pass
#endif
    const int keep_skip = c->kd.skip_prep;
    c->kd.aux = 1;
    // The kd query of the agents this pass's grid query will list: answered by that query's own last workgroup, once this build's last
    // kernel has published its tree (KdTail), while the list lengths that have come back say "nobody" -- otherwise a launch of
    // its own behind the build (launch_policy).  The build belongs to the pass whose grid query comes next: auto_seq + 1, whether it is
    // enqueued inside that pass or ahead of it (sca_run_steps) -- in front of that grid query in the host's order either way.
    const unsigned seq = c->auto_seq + 1;
    c->kd_tail_arg = KdTail{};
    c->kd_block_stop = nullptr;
    if (c->auto_tail_ok && c->auto_waitvalue && seq != 0 && c->kdq_last >= 0 && c->kdq_last <= c->auto_tail_max) {
        // (the list of this parity was last read by the kd query of two passes ago, whose event slot is about to be recorded anew: the
        // pass's stream waits for it HERE instead of in front of its grid build)
        // (... if that query was a launch on kd_stream; in the launch-free form it ran inside the pass's own grid query, on this stream)
        if (seq >= 3 && c->kdq_on_kd_stream[(seq - 2) & 3u]) { if (int r = wait_if_pending(c, c->stream, c->ev_auto_kdq[(seq - 2) & 3u])) return r; }
        KdTail T;
        T.seq = seq; T.sync = c->auto_sync;
        c->kd_tail_arg = T;
        c->kd_tail_seq = seq;
        c->kd_block_stop = c->ev_auto_kdq[seq & 3u];
    }
    const int r = build_agent_tree_device(c, c->kd_stream, v);
    c->kd_tail_arg = KdTail{}; c->kd_block_stop = nullptr;
    c->kd.aux = 0; c->kd.skip_prep = keep_skip;
    return r;
}

// everything kd_stream was given has to be through before a caller reads the tree, the permutation or the lists, and before a pass that
// is not an AUTO pass builds in place: the context's stream waits for the last kd query
static int api_enter(sca_ctx *c) {
    if (!c->lazy_join) return 0;
    c->lazy_join = false;
    return auto_join(c);
}
static int auto_join(sca_ctx *c) {
    c->lazy_join = false;
    if (!c->auto_unjoined) return 0;
    CHK(c, hipStreamWaitEvent(c->stream, c->ev_auto_kd, 0));
    c->auto_unjoined = false;
    return 0;
}
// will the next pass of an SCA_NBR_AUTO run be an AUTO pass (and not a plain kd pass)?  Decides whether its tree may be built ahead.
static bool auto_next(const sca_ctx *c) {
    return c->auto_fits && c->auto_backoff == 0 && !c->part_on && !(c->trk_on && c->trk_in_pass) &&
           !(c->kdq_last >= 0 && (long long)c->kdq_last * c->auto_div > (long long)c->d.shard_count);
}

static int launch_policy(sca_ctx *c, int mode, bool timed, bool fuse_integrate) {
    const bool fork_ready = c->fork_ready;
    c->fork_ready = false;
#ifdef SCA_TIMELINE
    c->d.tl_step++;
#endif
    // SCA_NBR_AUTO resolves to a plain kd pass where it cannot help (the cell-owner partition has its own structures; the grid's
    // candidate lists need the collision reach inside one cell) or where the grid keeps listing a large part of the swarm for the kd
    // query anyway (a lattice of identical cells: ties everywhere) -- it is tried again every 256 passes
    bool auto_mode = false;
    if (mode == SCA_NBR_AUTO) {
        double ar, orr;
        collide_reach(c, ar, orr);
        const bool fits = c->max_radius + ar <= c->P.neighbor_dist && c->max_radius + orr <= c->P.neighbor_dist;
        if (c->part_on) { c->err = "the cell-owner partition is a mode of SCA_NBR_GRID"; return SCA_ERR_UNSUPPORTED; }
        if (int r = auto_prepare(c)) return r;
        if (c->kdq_pending && hipEventQuery(c->ev_auto_cnt) == hipSuccess) { c->kdq_last = c->kdq_host[0]; c->kdq_pending = false; }
        // (a tree built ahead for this pass -- sca_run_steps, see there -- makes it an AUTO pass whatever the counts say: the build
        // must not run twice; sca_run_steps only builds ahead when auto_next() holds)
        if (!c->kd_ahead && c->auto_backoff == 0 && c->kdq_last >= 0 && (long long)c->kdq_last * c->auto_div > (long long)c->d.shard_count) { c->auto_backoff = 256; c->kdq_last = -1; }
        // (a pass with the tracker inside runs its whole neighbour branch beside the re-plans already: nothing to gain, a grid build to lose)
        const bool tracked_pass = c->trk_on && c->trk_in_pass;
        if (!c->kd_ahead && (!fits || tracked_pass || c->auto_backoff > 0)) {
            if (c->auto_backoff > 0) c->auto_backoff--;
            mode = SCA_NBR_KDTREE;
        } else auto_mode = true;
        c->auto_fits = fits;
    }
    const bool kd_prebuilt = auto_mode && c->kd_ahead;
    c->kd_ahead = false;
    c->auto_ran = auto_mode;
    if (!auto_mode) { if (int r = auto_join(c)) return r; }            // (a pass that builds its tree in place: kd_stream must be through)
    int lp_lo = 0, lp_hi = 0;
    choose_lp_form(c, lp_lo, lp_hi);
    const DeviceView &d = c->d;
    const int32_t *lp_ids = c->part_on ? d.own : c->lp_list;
    // the tracker's re-plans overlap the device kd build and the neighbour query; the per-agent prologue (which reads
    // v_pref) of the tracker's agents then moves from k_kd_gather to the tracker's kernels (track_store)
    const bool tracked = c->trk_on && c->trk_in_pass;
    const bool overlap = tracked && (mode == SCA_NBR_KDTREE || mode == SCA_NBR_GRID || auto_mode) && !c->trk_serial;
    // k_solve's v_pref-independent half right behind the neighbour query, i.e. beside the re-plans when they are overlapped
    bool split = choose_solve_split(c, overlap, d.shard_count);
    // k_solve that finishes its own fallbacks (k_solve_fb): while all the shard's wavefronts are resident at once even at the fallback
    // sweep's 252 registers (two per SIMD), and nobody else feeds the fallback list (k_lp does)
    const bool solve_fb = !split && lp_hi == lp_lo && !c->part_on && d.shard_count <= c->solve_fb_max;
    if (split && !c->d.sw_slot) {
        // scratch of the two-launch solve (cones + survivor lists, ~2 KB per agent): all three buffers or none, and a pass that
        // cannot have them runs the one-launch k_solve instead of failing
        const size_t N = (size_t)c->max_n;
        double *a = nullptr; uint16_t *b = nullptr; int32_t *e = nullptr;
        const bool ok = hipMalloc((void **)&a, sizeof(double) * N * K_MAX * SLOTF) == hipSuccess &&
                        hipMalloc((void **)&b, sizeof(uint16_t) * N * 512) == hipSuccess &&
                        hipMalloc((void **)&e, sizeof(int32_t) * N) == hipSuccess;
        if (ok) { c->d.sw_slot = a; c->d.sw_surv = b; c->d.sw_n = e; }
        else { (void)hipGetLastError(); if (a) (void)hipFree(a); if (b) (void)hipFree(b); if (e) (void)hipFree(e); split = false; }
    }
    c->forms = (split ? SCA_FORM_SOLVE_SPLIT : 0);
    if (split && lp_hi > lp_lo) c->d.lp_kernel = 1;                     // k_solve_pick4 carries no LP: its agents go to k_lp
    c->kd.skip_prep = overlap ? 1 : 0;
    c->grid.skip_prep = overlap ? 1 : 0;
    if (mode == SCA_NBR_GRID) {
        // K4's candidate lists and its fallback cover one cell around the agent: the collision reach must fit into it
        double agent_reach, obs_reach;
        collide_reach(c, agent_reach, obs_reach);
        if (c->max_radius + agent_reach > c->P.neighbor_dist || c->max_radius + obs_reach > c->P.neighbor_dist) {
            c->err = "SCA_NBR_GRID needs radius + collision reach <= neighbor_dist"; return SCA_ERR_UNSUPPORTED;
        }
    }
    // Streams.  Without the tracker everything is one chain on the main stream.  With it (overlap): the pass's critical path
    // k_track -> re-plans -> prologue -> pick stays on the main stream, and the neighbour structure + query (+ the sweep half of
    // k_solve), which depend on nothing of the tracker -- it reads agent.neighbors[0] from what the previous pass's epilogue
    // saved, not from the lists K1 overwrites -- run beside them on trk_stream:
    //   main: [fork] k_track re-plans (or k_track_replan) ........ [wait join] k_solve / k_solve_pick4 ...
    //   side: [wait fork] K0 ...... K1 (k_solve_sweep) [join]
    // (round 1 had the re-plans on the side stream: the fork and the join then sat on the critical path, ~40 us per step)
    c->nbr_stream = overlap ? c->trk_stream : c->stream;
    const unsigned parity_now = (unsigned)c->trk.parity;
    if (overlap) {
        if (!fork_ready) CHK(c, hipEventRecord(c->trk_fork, c->stream));   // (fork_ready: it rode on the previous step's last kernel, sca_run_steps)
        CHK(c, hipStreamWaitEvent(c->trk_stream, c->trk_fork, 0));
    }
    if (auto_mode && !kd_prebuilt) {
        // SCA_NBR_AUTO: the kd build (its permutation is history: every step) on a stream of its own, from the pass's start
        CHK(c, hipEventRecord(c->ev_auto_fork, c->stream));
        if (int r = auto_enqueue_kd_build(c, c->ev_auto_fork, c->d.rec)) return r;
    }
    c->trk.prep = overlap ? 1 : 0;
    c->trk.P = c->P;
    if (tracked) { if (int r = launch_tracker(c, true, true)) return r; }
    c->nbr_mode = auto_mode ? (int)SCA_NBR_GRID : mode;               // (what K4's fallback looks into: the grid in an AUTO pass)
    if (auto_mode) {
        c->d.kdq_cap = std::max(1, c->d.shard_count / c->auto_div);
        {   // this pass's list (the previous pass's kd query may still be looking at the other one's length)
            const unsigned par = (c->auto_seq + 1) & 1u;
            c->d.kdq_list = c->kdq_list + (size_t)par * c->max_n;
            c->d.kdq_count = c->kdq_count + par;
            // ... which the kd query of two passes ago must be through with (it is, unless the kd stream lags by two whole passes)
            // (slot [(seq - 2) & 3], seq = auto_seq + 1; a build enqueued in the tail form has made the pass's stream wait already)
            if (c->auto_seq >= 2 && c->kd_tail_seq != c->auto_seq + 1 && c->kdq_on_kd_stream[(c->auto_seq - 1) & 3u]) { if (int r = wait_if_pending(c, c->nbr_stream, c->ev_auto_kdq[(c->auto_seq - 1) & 3u])) return r; }
        }
        if (int r = build_agent_grid_device(c)) return r;
    }
    else if (mode == SCA_NBR_KDTREE) { if (int r = build_agent_tree_device(c, c->nbr_stream, c->d)) return r; }
    else if (mode == SCA_NBR_GRID) { if (int r = build_agent_grid_device(c)) return r; }
    else if (mode == SCA_NBR_KDTREE_HOSTBUILD) {
        if (c->perm_on_device) {
            CHK(c, hipMemcpyAsync(c->h_perm.data(), c->d.aperm, sizeof(int32_t) * c->n, hipMemcpyDeviceToHost, c->stream));
            CHK(c, hipStreamSynchronize(c->stream));
            c->perm_on_device = false;
        }
        if (int r = build_agent_tree(c)) return r;
        hipLaunchKernelGGL(k_prep, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->d, c->P);
    } else { c->err = "neighbor mode not available in this build"; return SCA_ERR_UNSUPPORTED; }
    const int cnt = d.shard_count;
    hipStream_t ns = c->nbr_stream;
    hipEvent_t e0 = c->ev[0], e1 = c->ev[1], e2 = c->ev[2], e3 = c->ev[3];
    const bool prof = !timed && c->profiling && (c->prof_tick++ & 15u) == 0 && c->pool_used + 4 <= 4 * 4096;
    if (prof) { if (pool_event(c, &e0) || pool_event(c, &e1) || pool_event(c, &e2) || pool_event(c, &e3)) return SCA_ERR_HIP; }
    if (timed || prof) CHK(c, hipEventRecord(e0, ns));
    double agent_reach, obs_reach;
    collide_reach(c, agent_reach, obs_reach);
    // packed K1 (4 agents per wavefront) wins once the shard fills the chip (measured: 2.3x at 16k agents, equal at 6000);
    // below that the one-agent-per-wave form with its record stack has the shorter critical path (4096 random: 30 % faster)
    const bool packed = c->k1_force < 0 ? cnt >= per_simd(c, 6144) : c->k1_force != 0;   // four agents per wavefront once that still fills the SIMDs
    // (overlapped: the join rides on the side stream's last kernel -- the neighbour query, or the sweep behind it)
    const hipEvent_t k1_stop = overlap && !split && c->ext_stop ? c->trk_join : nullptr;
    const hipEvent_t sweep_stop = overlap && split && c->ext_stop ? c->trk_join : nullptr;
    const bool obs = d.m > 0;                                           // (scenes without obstacles: the K1 forms without the obstacle phase, see neighbors_one)
    if (auto_mode) {
        const int per_block = K1P_WAVES * K1P_APW;
        // (the launch-free form: the grid query's last workgroup answers whoever is listed, from the tree its pass's build publishes; d is c->d)
        c->d.auto_sync = c->kd_tail_seq == c->auto_seq + 1 && c->kd_tail_seq != 0 ? c->auto_sync : nullptr;
        c->d.auto_pass_seq = c->auto_seq + 1;
        c->d.auto_err = c->kd.counts + KD_MAX_LEVELS + 1;
        if (obs) LAUNCH_REC(c, c->ev_auto_k1g, (k_neighbors_grid<true, true>), dim3((cnt + per_block - 1) / per_block), dim3(K1P_WAVES * 64), ns, d, c->grid,
                            c->P, agent_reach, obs_reach, c->max_radius);
        else LAUNCH_REC(c, c->ev_auto_k1g, (k_neighbors_grid<true, false>), dim3((cnt + per_block - 1) / per_block), dim3(K1P_WAVES * 64), ns, d, c->grid,
                        c->P, agent_reach, obs_reach, c->max_radius);
    } else if (mode == SCA_NBR_GRID) {
        const int per_block = K1P_WAVES * K1P_APW;
        if (obs) LAUNCH_OPT(c, k1_stop, (k_neighbors_grid<false, true>), dim3((cnt + per_block - 1) / per_block), dim3(K1P_WAVES * 64), ns, d, c->grid,
                            c->P, agent_reach, obs_reach, c->max_radius);
        else LAUNCH_OPT(c, k1_stop, (k_neighbors_grid<false, false>), dim3((cnt + per_block - 1) / per_block), dim3(K1P_WAVES * 64), ns, d, c->grid,
                        c->P, agent_reach, obs_reach, c->max_radius);
    } else if (packed) {
        const int per_block = K1P_WAVES * K1P_APW;
        if (obs) LAUNCH_OPT(c, k1_stop, k_neighbors_kd4<true>, dim3((cnt + per_block - 1) / per_block), dim3(K1P_WAVES * 64), ns, d, c->P,
                            agent_reach, obs_reach, c->max_radius);
        else LAUNCH_OPT(c, k1_stop, k_neighbors_kd4<false>, dim3((cnt + per_block - 1) / per_block), dim3(K1P_WAVES * 64), ns, d, c->P,
                        agent_reach, obs_reach, c->max_radius);
    } else
        if (obs) LAUNCH_OPT(c, k1_stop, k_neighbors_kd<true>, dim3(std::min((cnt + K1_WAVES - 1) / K1_WAVES, MAX_GRID)), dim3(K1_WAVES * 64),
                            ns, d, c->P, agent_reach, obs_reach, c->max_radius);
        else LAUNCH_OPT(c, k1_stop, k_neighbors_kd<false>, dim3(std::min((cnt + K1_WAVES - 1) / K1_WAVES, MAX_GRID)), dim3(K1_WAVES * 64),
                        ns, d, c->P, agent_reach, obs_reach, c->max_radius);
    c->near_valid = true;
    if ((timed || prof) && !auto_mode) CHK(c, hipEventRecord(e1, ns));    // [e0, e1] = K1
    if (auto_mode) {
        // the agents the grid query listed (more than max_neighbors in range, equal distances) get the kd-tree's answer: behind the
        // build on kd_stream, and nothing reads a list before that query is through
        const unsigned seq = ++c->auto_seq;                            // (its parity picks the list; it may wrap)
        if (c->kd_tail_seq == seq && seq != 0) {
            // the launch-free form: the grid query's own last workgroup has answered the listed agents from this pass's tree (KdTail) -- no
            // launch, no cross-stream wait on the build's stream, no stream wait operation on this one
            c->forms |= SCA_FORM_AUTO_TAIL;
            c->kdq_on_kd_stream[seq & 3u] = false;
        } else {
            c->kdq_on_kd_stream[seq & 3u] = true;
            CHK(c, hipStreamWaitEvent(c->kd_stream, c->ev_auto_k1g, 0));     // (recorded behind the grid query above)
            // (ev_auto_kdq: also "the last kd query": auto_join and the event form of the wait)
            LAUNCH_REC(c, c->ev_auto_kdq[seq & 3u], k_neighbors_kd_auto, dim3(c->kdq_last >= 0 && c->kdq_last <= KDQ_BLOCKS_FEW * K1_WAVES ? KDQ_BLOCKS_FEW : KDQ_BLOCKS),
                       dim3(K1_WAVES * 64), c->kd_stream, d, c->P, agent_reach, obs_reach, c->max_radius, c->auto_ticket);
        }
        if (!c->kdq_pending && (c->auto_passes++ & 3u) == 0) {       // how many were listed: for later passes' choice, never waited for
            CHK(c, hipMemcpyAsync(c->kdq_host, d.kdq_count, sizeof(int), hipMemcpyDeviceToHost, c->kd_stream));
            CHK(c, hipEventRecord(c->ev_auto_cnt, c->kd_stream));
            c->kdq_pending = true;
        }
        c->ev_auto_kd = c->ev_auto_kdq[seq & 3u];
        if (c->forms & SCA_FORM_AUTO_TAIL) {
            // (the launch-free form: the lists are final when the grid query's launch ends -- its last workgroup answered whoever was listed)
        } else if (c->auto_waitvalue) {
            // lists final: at once when the grid query listed nobody, else behind the kd query (see k_neighbors_kd_auto)
            if (hipStreamWaitValue32(ns, c->auto_busy, 0u, hipStreamWaitValueEq, 1u) != hipSuccess) {
                (void)hipGetLastError();                              // a platform without stream memory operations: the event wait from now on
                c->auto_waitvalue = false;
                CHK(c, hipStreamWaitEvent(ns, c->ev_auto_kd, 0));
            }
        } else CHK(c, hipStreamWaitEvent(ns, c->ev_auto_kd, 0));
        if (timed || prof) CHK(c, hipEventRecord(e1, ns));               // [e0, e1] of an AUTO pass: the grid query AND the wait for the kd query
                                                                        // of the listed agents behind it (the lists are final here)
    }
    if (split) {
        LAUNCH_OPT(c, sweep_stop, k_solve_sweep, dim3((cnt + SOLVE_WAVES - 1) / SOLVE_WAVES), dim3(SOLVE_WAVES * 64), ns, d, c->P);
    }
    if (overlap) {
        // the re-plan count of this pass for a later pass's launch decision: on the side stream, every 4th pass, never waited for
        const unsigned pass = c->trk_passes++;
        if (!c->trk_count_pending && (pass & 3u) == 1) {                    // the previous pass's slot: final since before the fork
                                                                            // (first at the second pass: the forms settle within three)
            CHK(c, hipMemcpyAsync(c->trk_host_count, c->trk.count + ((parity_now + 3u) & 3u), sizeof(int), hipMemcpyDeviceToHost, ns));
            CHK(c, hipEventRecord(c->trk_count_ev, ns));
            c->trk_count_pending = true;
        }
        if (!k1_stop && !sweep_stop) CHK(c, hipEventRecord(c->trk_join, ns));
        CHK(c, hipStreamWaitEvent(c->stream, c->trk_join, 0));                // (the prologue: track_store / the gather)
    }
    if (timed || prof) CHK(c, hipEventRecord(e2, c->stream));
    if (split) {
        const int per_block = SOLVE_WAVES * PICK_APW;
        hipLaunchKernelGGL(k_solve_pick4, dim3((cnt + per_block - 1) / per_block), dim3(SOLVE_WAVES * 64), 0, c->stream, d, c->P);
    }
    else if (solve_fb) {
        hipLaunchKernelGGL(k_solve_fb, dim3((cnt + SOLVE_WAVES - 1) / SOLVE_WAVES), dim3(SOLVE_WAVES * 64), 0, c->stream, d, c->P);
        c->forms |= SCA_FORM_SOLVE_FB;
    }
    else {
        hipLaunchKernelGGL(k_solve, dim3((cnt + SOLVE_WAVES - 1) / SOLVE_WAVES), dim3(SOLVE_WAVES * 64), 0, c->stream, d, c->P);
        if (!d.lp_kernel && lp_hi > lp_lo)                            // K3, one wavefront per LP agent (few of them)
            hipLaunchKernelGGL(k_solve_lpw, dim3((lp_hi - lp_lo + SOLVE_WAVES - 1) / SOLVE_WAVES), dim3(SOLVE_WAVES * 64), 0, c->stream, d, c->P,
                               lp_ids, lp_lo, lp_hi);
    }
    if (d.lp_kernel) c->forms |= SCA_FORM_LP_LANE;
    if (d.lp_kernel)                                                  // K3: the LP agents of the shard, one lane each
        hipLaunchKernelGGL(k_lp, dim3((lp_hi - lp_lo + 63) / 64), dim3(64), 0, c->stream, d, c->P, lp_ids, lp_lo, lp_hi);
    if (timed || prof) CHK(c, hipEventRecord(e3, c->stream));         // [e2, e3] = k_solve (+ k_lp) (what rocprofv3 reports for them)
    // the agents without any suitable candidate (rare; one wavefront each), then the epilogue (one lane per agent)
    const int ablocks = (cnt + 255) / 256;
    if (auto_mode && c->auto_waitvalue) {
        // nothing of this pass waited for the kd build any more.  Two things still must: (i) the integrate stage writes the record
        // buffer that the build BEFORE this pass's read its positions from (the two buffers alternate): wait for that build's gather;
        if (int r = wait_if_pending(c, c->stream, c->ev_auto_gather[c->auto_builds & 1u])) return r;       // [builds & 1] = the one before the last
        c->auto_unjoined = true;                                        // (ii) see auto_join
    }
    // (small shards: the fallback sweep rides in the epilogue's launch -- k_action_fb -- instead of in front of it)
    const bool action_fb = !solve_fb && cnt <= c->action_fb_max;
    if (action_fb) c->forms |= SCA_FORM_ACTION_FB;
    if (fuse_integrate) {
        if (action_fb) LAUNCH_OPT(c, c->action_stop, k_action_fb<true>, dim3(ablocks + FB_BLOCKS_SMALL), dim3(SOLVE_WAVES * 64), c->stream, d, c->P, ablocks);
        else {
            if (!solve_fb) hipLaunchKernelGGL(k_fallback<true>, dim3(FB_BLOCKS), dim3(SOLVE_WAVES * 64), 0, c->stream, d, c->P);
            LAUNCH_OPT(c, c->action_stop, k_action<true>, dim3(ablocks), dim3(256), c->stream, d, c->P);
        }
    } else if (action_fb) hipLaunchKernelGGL(k_action_fb<false>, dim3(ablocks + FB_BLOCKS_SMALL), dim3(SOLVE_WAVES * 64), 0, c->stream, d, c->P, ablocks);
    else {
        if (!solve_fb) hipLaunchKernelGGL(k_fallback<false>, dim3(FB_BLOCKS), dim3(SOLVE_WAVES * 64), 0, c->stream, d, c->P);
        hipLaunchKernelGGL(k_action<false>, dim3(ablocks), dim3(256), 0, c->stream, d, c->P);
    }
    CHK(c, hipGetLastError());
    if (fuse_integrate && c->d.hist) c->d.hist_row++;
    return 0;
}

static int launch_integrate(sca_ctx *c) {
    const DeviceView &d = c->d;
    const int cnt = d.shard_count;
    hipLaunchKernelGGL(k_integrate, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, d, c->P);
    CHK(c, hipGetLastError());
    if (c->d.hist) c->d.hist_row++;
    return 0;
}
// check_agent_state + is_done; afterwards the moved records become the current ones (buffer swap, no copy)
static int launch_collide_finish(sca_ctx *c, bool timed) {
    DeviceView &d = c->d;
    const int cnt = d.shard_count;
    double agent_reach, obs_reach;
    collide_reach(c, agent_reach, obs_reach);
    if (!c->near_valid) CHK(c, hipMemsetAsync(d.done_count, 0, sizeof(int32_t) * 256 * 32, c->stream));   // no policy pass before
    if (!c->near_valid) hipLaunchKernelGGL(k_invalidate_near, dim3((d.n + 255) / 256), dim3(256), 0, c->stream, d);
    if (!c->near_valid && c->nbr_mode == SCA_NBR_GRID) {                  // the fallback reads the grid: make it describe these records
        c->grid.skip_prep = 2;
        c->nbr_stream = c->stream;
        if (int r = build_agent_grid_device(c)) return r;
    }
    c->near_valid = false;
    const dim3 k4grid((cnt + K4_WAVES * K4_APW - 1) / (K4_WAVES * K4_APW));
    // (the event that rides on the step's last kernel, if sca_run_steps asked for one: the next pass's fork)
    const bool others = c->part_on || cnt < d.n;
    const hipEvent_t k4_stop = others ? nullptr : c->finish_stop, others_stop = others ? c->finish_stop : nullptr;
    const int fresh = c->state_fresh ? 1 : 0;
    if (c->nbr_mode == SCA_NBR_GRID)
        LAUNCH_OPT(c, k4_stop, k_collide_finish_grid, k4grid, dim3(K4_WAVES * 64), c->stream, d, c->grid, c->P, agent_reach, obs_reach, fresh);
    else
        LAUNCH_OPT(c, k4_stop, k_collide_finish, k4grid, dim3(K4_WAVES * 64), c->stream, d, c->P, agent_reach, obs_reach, fresh);
    c->state_fresh = false;
    if (c->part_on) {
        // sized with the whole present bound: d.shard_count and d.n_present are the host's UPPER bounds (part_bounds clamps the halo
        // bound to n - owned bound, i.e. to 0 for small swarms), so their difference says nothing about the halo count -- the kernel
        // reads the exact counts and returns beyond them (halo copies that arrived at their goal kept flying for their
        // neighbours on this rank for one step)
        LAUNCH_OPT(c, others_stop, k_goal_flags_others, dim3((std::max(1, d.n_present) + 255) / 256), dim3(256), c->stream, d, c->P);
    } else if (cnt < d.n) LAUNCH_OPT(c, others_stop, k_goal_flags_others, dim3((d.n + 255) / 256), dim3(256), c->stream, d, c->P);
    if (timed) CHK(c, hipEventRecord(c->ev[5], c->stream));
    CHK(c, hipGetLastError());
    std::swap(d.rec, d.rec_new);
    c->h_pos_valid = false;
    return 0;
}
static int launch_update(sca_ctx *c, bool timed) {
    if (int r = launch_integrate(c)) return r;
    return launch_collide_finish(c, timed);
}

int sca_policy_pass(sca_ctx *c, int neighbor_mode) {
    API_ENTER(c);
    if (!c->state_set) { c->err = "sca_set_state first"; return SCA_ERR_STATE; }
    if (int r = launch_policy(c, neighbor_mode, true, false)) return r;
    if (int r = auto_join(c)) return r;
    CHK(c, hipStreamSynchronize(c->stream));
    if (neighbor_mode == SCA_NBR_KDTREE || neighbor_mode == SCA_NBR_AUTO) { if (int r = check_kd_overflow(c)) return r; }
    CHK(c, hipEventElapsedTime(&c->ms_nbr, c->ev[0], c->ev[1]));
    CHK(c, hipEventElapsedTime(&c->ms_solve, c->ev[2], c->ev[3]));
    return 0;
}

static int read_active(sca_ctx *c, int *active, bool kd_word);
int sca_env_update(sca_ctx *c, int *all_done) {
    API_ENTER(c);
    if (!c->state_set) { c->err = "sca_set_state first"; return SCA_ERR_STATE; }
    CHK(c, hipEventRecord(c->ev[4], c->stream));
    if (int r = launch_update(c, true)) return r;
    if (all_done) {
        int active = 0;
        if (int r = read_active(c, &active, false)) return r;
        CHK(c, hipEventElapsedTime(&c->ms_update, c->ev[4], c->ev[5]));
        *all_done = (active == 0);
    }
    return 0;
}

// K4's counters (and the kd build's error word) in ONE round trip through a pinned buffer: this is the per-step synchronisation of the
// drop-in loop `while not env.step()` (round 4: two pageable copies with a stream synchronisation each)
static int read_active(sca_ctx *c, int *active, bool kd_word) {
    constexpr int PARTS = 256 * 32;
    if (!c->h_done) CHK(c, hipHostMalloc((void **)&c->h_done, sizeof(int32_t) * (PARTS + 1)));
    c->h_done[PARTS] = 0;
    CHK(c, hipMemcpyAsync(c->h_done, c->d.done_count, sizeof(int32_t) * PARTS, hipMemcpyDeviceToHost, c->stream));
    if (kd_word) CHK(c, hipMemcpyAsync(c->h_done + PARTS, c->kd.counts + KD_MAX_LEVELS + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    if (c->h_done[PARTS]) return check_kd_overflow(c);                      // (reports, resets the word)
    int n = 0;
    for (int i = 0; i < PARTS; i += 32) n += c->h_done[i];
    *active = n;
    return 0;
}
int sca_active_count(sca_ctx *c, int *active) {
    API_ENTER(c);
    ARG(c, active);
    if (!c->state_set) { c->err = "sca_set_state first"; return SCA_ERR_STATE; }
    return read_active(c, active, c->perm_on_device);
}

// the step's exchange: every rank contributes its shard's moved records, in place
static int exchange_moved_records(sca_ctx *c) {
    const size_t bytes = sizeof(PubRec) * (size_t)c->d.shard_count;
    hipEvent_t x0 = nullptr, x1 = nullptr;
    if (c->profiling && c->pool_xch_used + 2 <= 2 * 4096) {              // the collective's device time, on the stream it is enqueued on
        for (hipEvent_t *ev : {&x0, &x1}) {
            if (c->pool_xch_used == (int)c->pool_xch.size()) { hipEvent_t n_; CHK(c, hipEventCreate(&n_)); c->pool_xch.push_back(n_); }
            *ev = c->pool_xch[c->pool_xch_used++];
        }
        CHK(c, hipEventRecord(x0, c->stream));
    }
    const ncclResult_t e = g_rccl.AllGather((const char *)c->d.rec_new + sizeof(PubRec) * (size_t)c->d.shard_begin, c->d.rec_new, bytes,
                                            ncclChar, c->comm, c->stream);
    if (e != ncclSuccess) { c->err = std::string("ncclAllGather: ") + g_rccl.GetErrorString(e); return SCA_ERR_HIP; }
    if (x1) CHK(c, hipEventRecord(x1, c->stream));
    return 0;
}

static int run_steps_loop(sca_ctx *c, int steps, int neighbor_mode, bool lazy);
// A failure in the middle of the loop (an exchange, a launch, a partition commit) must not leave the AUTO bookkeeping pointing at a tree
// built ahead from positions the caller is about to replace: forget the build-ahead, let the context's stream wait for whatever
// kd_stream still holds, and have the next AUTO pass decide afresh.
static void auto_abandon(sca_ctx *c) {
    c->lazy_join = false;
    c->kd_ahead = false; c->kdq_last = -1; c->kdq_pending = false; c->auto_backoff = 0;
    c->kd_tail_seq = 0;
    for (bool &b : c->kdq_on_kd_stream) b = false;
    if (c->kd_stream) { (void)hipStreamSynchronize(c->kd_stream); (void)hipGetLastError(); }
    if (c->auto_sync) {
        // (tickets of a pass that was cut short: the words start from zero again)
        (void)hipStreamSynchronize(c->stream);
        (void)hipMemset(c->auto_sync, 0, 8 * sizeof(unsigned));
        (void)hipGetLastError();
    }
    c->auto_unjoined = false;
}
static int run_steps_guarded(sca_ctx *c, int steps, int neighbor_mode, bool lazy) {
    const int r = run_steps_loop(c, steps, neighbor_mode, lazy);
    if (r != 0) { const std::string keep = c->err; auto_abandon(c); c->err = keep; }
    return r;
}
int sca_run_steps(sca_ctx *c, int steps, int neighbor_mode) {
    API_ENTER(c);
    return run_steps_guarded(c, steps, neighbor_mode, false);
}
static int run_steps_loop(sca_ctx *c, int steps, int neighbor_mode, bool lazy) {
    c->fork_ready = false; c->finish_stop = nullptr; c->action_stop = nullptr;
    if (!c->state_set) { c->err = "sca_set_state first"; return SCA_ERR_STATE; }
    if (steps < 0) { c->err = "steps must not be negative"; return SCA_ERR_ARG; }
    if (neighbor_mode < SCA_NBR_KDTREE || neighbor_mode > SCA_NBR_AUTO) {     // (also for steps == 0: a wrong mode is a wrong call)
        c->err = "neighbor mode not available in this build"; return SCA_ERR_UNSUPPORTED;
    }
    if (c->part_on && c->part_nranks > 1 && !c->shard_emulation) {
        c->err = "cell-owner partition over several ranks: drive the step with sca_step_begin / sca_partition_pack / [exchange] / "
                 "sca_partition_unpack / sca_partition_commit / sca_step_end (sca_amd.distributed.PartitionedStepper)";
        return SCA_ERR_STATE;
    }
    if (c->part_on && c->part_nranks > 1) {                                 // emulation: this rank alone, its messages go nowhere
        for (int k = 0; k < 2; k++)
            if (!c->part_scratch[k]) CHK(c, hipMalloc((void **)&c->part_scratch[k], part_message_bytes(c->part.cap_halo, c->part.cap_mig)));
    }
    if (c->part_on && neighbor_mode != SCA_NBR_GRID) { c->err = "the cell-owner partition is a mode of SCA_NBR_GRID"; return SCA_ERR_UNSUPPORTED; }
    for (int s = 0; s < steps; s++) {
        // (SCA_NBR_AUTO, another step to follow: the moved positions' event rides on k_action -- the next pass's kd build waits on it)
        c->action_stop = neighbor_mode == SCA_NBR_AUTO && s + 1 < steps && c->kd_stream && c->ext_stop && !c->comm && c->d.shard_count == c->n
                             ? c->ev_auto_moved : nullptr;                      // (a shard: the records of the others arrive after k_action)
        const int rp = launch_policy(c, neighbor_mode, false, true);           // integrate fused into k_solve
        const bool moved_recorded = c->action_stop != nullptr;
        c->action_stop = nullptr;
        if (rp) return rp;
        if (c->part_on) {                                                      // one rank (or one rank alone, emulation): nobody to exchange with
            if (c->part_nranks > 1) { if (int r = sca_partition_pack(c, c->part_scratch[0], c->part_scratch[1])) return r; }
            if (int r = sca_partition_commit(c)) return r;
        }
        if (c->comm) { if (int r = exchange_moved_records(c)) return r; }
        else if (c->shard_emulation && !c->part_on && c->d.shard_count < c->n) {      // (partition: the halo copies stand still by themselves)
            // stand-in for the all-gather's arrivals: the other ranks' agents stand still (their records are copied over)
            const int b = c->d.shard_begin, e = b + c->d.shard_count;
            if (b > 0) CHK(c, hipMemcpyAsync(c->d.rec_new, c->d.rec, sizeof(PubRec) * (size_t)b, hipMemcpyDeviceToDevice, c->stream));
            if (e < c->n) CHK(c, hipMemcpyAsync(c->d.rec_new + e, c->d.rec + e, sizeof(PubRec) * (size_t)(c->n - e), hipMemcpyDeviceToDevice, c->stream));
        }
        if (neighbor_mode == SCA_NBR_AUTO && c->auto_ran && s + 1 < steps && auto_next(c)) {
            // SCA_NBR_AUTO: the NEXT pass's kd build needs the moved positions only -- k_collide_finish changes flags, and in an
            // AUTO pass its fallback looks into the grid, not into the tree -- so it starts here, beside the collision check and the
            // next pass's grid build and query.  Only inside one call: a caller who reads the permutation between calls sees as many
            // builds as steps.
            // (the collision check goes to its stream FIRST: the host needs ~5 us per launch, and the build's eight or more launches in
            // front of it left the main stream idle for 50 us at N = 4096)
            if (!moved_recorded) CHK(c, hipEventRecord(c->ev_auto_moved, c->stream));
            if (int r = launch_collide_finish(c, false)) return r;                 // ((no tracker in an AUTO pass: no fork to carry) swaps the record buffers: the moved ones are c->d.rec now)
            c->kd_ahead_enqueue = true;
            const int rb = auto_enqueue_kd_build(c, c->ev_auto_moved, c->d.rec);
            c->kd_ahead_enqueue = false;
            if (rb) return rb;
            c->kd_ahead = true;
            continue;
        }
        // the next pass's fork (tracker in the pass, its neighbour branch on the side stream) rides on this step's last kernel
        c->finish_stop = s + 1 < steps && c->ext_stop && c->trk_on && c->trk_in_pass && !c->trk_serial && c->trk_fork ? c->trk_fork : nullptr;
        const int rf = launch_collide_finish(c, false);
        c->fork_ready = rf == 0 && c->finish_stop != nullptr;
        c->finish_stop = nullptr;
        if (rf) return rf;
    }
    if (lazy && c->auto_unjoined && c->auto_ran) { c->lazy_join = true; return 0; }   // (sca_env_step: see API_ENTER)
    return auto_join(c);
}
int sca_env_step(sca_ctx *c, int neighbor_mode, int *active) {
    if (!c) return SCA_ERR_ARG;                                           // (not API_ENTER: a step after a step needs no join)
    ARG(c, active);
    if (int r = run_steps_guarded(c, 1, neighbor_mode, true)) return r;
    // (the kd build's error word is read from the context's stream: a build still running on kd_stream is seen one step later)
    return read_active(c, active, c->perm_on_device);
}
int sca_set_shard_emulation(sca_ctx *c, int on) {
    API_ENTER(c);
    if (c->comm && on) { c->err = "sca_set_shard_emulation with an active communicator: the exchange is real"; return SCA_ERR_STATE; }
    c->shard_emulation = on != 0;
    return 0;
}
int sca_last_pass_forms(sca_ctx *c, int *forms) {
    API_ENTER(c);
    ARG(c, forms);
    *forms = c->forms;
    return 0;
}
int sca_last_kd_build_ms(sca_ctx *c, float *kd_build_ms) {
    API_ENTER(c);
    ARG(c, kd_build_ms);
    *kd_build_ms = c->ms_kd_build;
    return 0;
}
int sca_auto_stats(sca_ctx *c, int64_t *out4, int reset) {
    API_ENTER(c);
    ARG(c, out4);
    for (int k = 0; k < 4; k++) out4[k] = 0;
    if (!c->d.kdq_stats) return 0;                                       // no SCA_NBR_AUTO pass has run
    if (c->kd_stream) CHK(c, hipStreamSynchronize(c->kd_stream));
    unsigned long long h[4];
    CHK(c, hipMemcpy(h, c->d.kdq_stats, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; k++) out4[k] = (int64_t)h[k];
    if (reset) CHK(c, hipMemset(c->d.kdq_stats, 0, sizeof(h)));
    return 0;
}
int sca_last_exchange_ms(sca_ctx *c, float *exchange_ms) {
    API_ENTER(c);
    ARG(c, exchange_ms);
    *exchange_ms = c->ms_exchange;
    return 0;
}
int sca_last_replan_ms(sca_ctx *c, float *replan_ms) {
    API_ENTER(c);
    if (replan_ms) *replan_ms = c->ms_replan;
    return 0;
}

// ---- RCCL inside the library (SURVEY.md 8b `sca_comm_init`, 8e) ---------------------------------------------------------
int sca_comm_probe(void) {
    return rccl_load() ? SCA_ERR_UNSUPPORTED : 0;
}
int sca_comm_unique_id(void *id_out) {
    if (!id_out) return SCA_ERR_ARG;
    if (rccl_load()) return SCA_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return SCA_ERR_HIP;
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}
int sca_comm_init(sca_ctx *c, int rank, int nranks, const void *unique_id) {
    API_ENTER(c);
    ARG(c, unique_id && nranks >= 1 && rank >= 0 && rank < nranks);
    if (!c->agents_set) { c->err = "sca_set_agents first"; return SCA_ERR_STATE; }
    if (c->comm) { c->err = "communicator already initialised (sca_comm_destroy first)"; return SCA_ERR_STATE; }
    if (c->part_on) { c->err = "sca_comm_init with the cell-owner partition active"; return SCA_ERR_STATE; }
    if (c->n % nranks) { c->err = "agent count must be a multiple of the rank count"; return SCA_ERR_ARG; }
    if (const char *e = rccl_load()) { c->err = e; return SCA_ERR_UNSUPPORTED; }
    CHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    const ncclResult_t e = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (e != ncclSuccess) { c->comm = nullptr; c->err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(e); return SCA_ERR_HIP; }
    c->comm_rank = rank; c->comm_nranks = nranks;
    c->d.shard_count = c->n / nranks; c->d.shard_begin = rank * c->d.shard_count;
    return 0;
}
int sca_comm_destroy(sca_ctx *c) {
    API_ENTER(c);
    if (c->comm) {
        CHK(c, hipStreamSynchronize(c->stream));
        (void)g_rccl.CommDestroy(c->comm);
        c->comm = nullptr; c->comm_rank = 0; c->comm_nranks = 1;
        c->d.shard_begin = 0; c->d.shard_count = c->n;
    }
    return 0;
}

// ---- cell-owner partition of SCA_NBR_GRID with halo exchange (sca_partition.hip.h) ---------------------
static void part_view(sca_ctx *c) {
    c->d.own = c->d.present = c->part.present[c->part.cur];
    c->d.count_dev = c->part.counts;
    c->d.shard_begin = 0;
    c->d.shard_count = c->part_counts[0];
    c->d.n_present = c->part_counts[0] + c->part_counts[1];
}
static int part_free(sca_ctx *c) {
    if (!c->part_on && !c->part.counts) return 0;
    CHK(c, hipStreamSynchronize(c->stream));
    for (void *q : {(void *)c->part.present[0], (void *)c->part.present[1], (void *)c->part.halo_tmp, (void *)c->part.counts, (void *)c->part.emig})
        if (q) (void)hipFree(q);
    if (c->part_host) { (void)hipHostFree(c->part_host); c->part_host = nullptr; }
    for (int k = 0; k < 2; k++) if (c->part_scratch[k]) { (void)hipFree(c->part_scratch[k]); c->part_scratch[k] = nullptr; }
    if (c->part_ev) { (void)hipEventDestroy(c->part_ev); c->part_ev = nullptr; }
    c->part = PartDev{};
    c->part_on = false; c->part_rank = 0; c->part_nranks = 1; c->part_pending = false;
    c->d.own = c->d.present = nullptr; c->d.count_dev = nullptr; c->d.n_present = 0;
    c->d.shard_begin = 0; c->d.shard_count = c->n;
    return 0;
}
// The launch bounds from what the host knows: the exact counts of a commit one or two steps back + room for what can have
// arrived since.  Agents move 0.1 m per step and cells are 10 m wide, so a step changes an owner's count by the few agents within
// 0.1 m of a cut; the margin is a sixteenth of the rank's agents (+ 4096) per commit of age -- and k_part_keep checks it: a launch
// that turned out too small sets an error bit instead of leaving agents out silently.
static void part_bounds(sca_ctx *c) {
    const long long grow = (long long)c->part_age + 1;
    c->part_counts[0] = (int)std::min<long long>(c->n, c->part_known[0] + grow * (c->part_known[0] / 16 + 4096));
    c->part_counts[1] = (int)std::min<long long>(c->n, c->part_known[1] + grow * (c->part_known[1] / 4 + 4096));
    if (c->part_counts[0] + c->part_counts[1] > c->n) c->part_counts[1] = c->n - c->part_counts[0];
}
static int part_report(sca_ctx *c, int bits) {
    if (!bits) return 0;
    c->err = bits & 4 ? "partition: owned + halo exceed the agent count (corrupt lists)"
           : bits & 8 ? "partition: a rank's agent count outgrew the launch bound between two read-backs (results of that step are incomplete)"
                      : "partition: a halo / migration message overflowed its capacity (sca_partition_init caps)";
    return SCA_ERR_STATE;
}
// lists being built -> current lists (behind k_part_close), without waiting for the device: the counts move on the device, the
// host asks for a copy and sizes its launches with bounds until it has arrived.  wait: get the exact counts now.
static int part_adopt(sca_ctx *c, bool wait) {
    c->part.cur ^= 1;
    c->part_age++;
    if (c->part_pending) c->part_copy_age++;
    if (c->part_pending && hipEventQuery(c->part_ev) == hipSuccess) {
        c->part_known[0] = c->part_host[0]; c->part_known[1] = c->part_host[1]; c->part_counts[4] = c->part_host[4];
        c->part_age = c->part_copy_age;                                     // commits since that snapshot
        c->part_pending = false;
    }
    if (!c->part_pending || wait) {
        if (c->part_pending) CHK(c, hipEventSynchronize(c->part_ev));
        CHK(c, hipMemcpyAsync(c->part_host, c->part.counts, sizeof(int) * 5, hipMemcpyDeviceToHost, c->stream));   // a snapshot behind this commit
        CHK(c, hipEventRecord(c->part_ev, c->stream));
        c->part_copy_age = 0;
        c->part_pending = true;
    }
    if (wait) {
        CHK(c, hipEventSynchronize(c->part_ev));
        c->part_known[0] = c->part_host[0]; c->part_known[1] = c->part_host[1]; c->part_counts[4] = c->part_host[4];
        c->part_age = 0; c->part_pending = false;
        c->part_counts[0] = c->part_known[0]; c->part_counts[1] = c->part_known[1];
    } else part_bounds(c);
    part_view(c);
    return part_report(c, c->part_counts[4]);
}
static int part_classify(sca_ctx *c) {                                      // from records every rank holds completely
    const int n = c->n;
    CHK(c, hipMemsetAsync(c->part.counts, 0, sizeof(int) * 16, c->stream));
    hipLaunchKernelGGL(k_part_init, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d, c->part);
    hipLaunchKernelGGL(k_part_init_halo, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d, c->part);
    hipLaunchKernelGGL(k_part_close<false>, dim3(64), dim3(256), 0, c->stream, c->d, c->part);
    CHK(c, hipGetLastError());
    CHK(c, hipMemcpyAsync(c->part.counts, c->part.counts + 2, sizeof(int) * 2, hipMemcpyDeviceToDevice, c->stream));   // current = built
    CHK(c, hipMemsetAsync(c->part.counts + 2, 0, sizeof(int) * 2, c->stream));
    return part_adopt(c, true);
}
int sca_partition_init(sca_ctx *c, int rank, int nranks, int axis, const double *cuts, int cap_halo, int cap_mig) {
    // (per-agent solver attributes -- sca_set_agent_params, like every static per-agent input indexed by GLOBAL id and replicated on every
    // rank -- need nothing of the messages: the slabs are cut in cells of the LARGEST neighborDist (c->P is the envelope), so one layer of
    // cells is a halo wide enough for every agent's own range, and the collision reach is the largest step's)
    API_ENTER(c);
    ARG(c, nranks >= 1 && rank >= 0 && rank < nranks && axis >= 0 && axis <= 2 && cap_halo >= 0 && cap_mig >= 0);
    if (!c->agents_set || !c->state_set) { c->err = "sca_set_agents and sca_set_state (the complete state, on every rank) first"; return SCA_ERR_STATE; }
    if (c->comm) { c->err = "sca_partition_init with an active communicator (the all-gather mode)"; return SCA_ERR_STATE; }
    if (int r = part_free(c)) return r;
    const int n = c->n;
    const double inv_cell = grid_inv_cell(c->P.neighbor_dist);
    // the cuts: given (coordinates along the axis, nranks - 1 of them, ascending), or equal shares of the agents as they stand
    // now; moved onto cell boundaries either way.  Identical on every rank: they all hold the same state.
    std::vector<long long> cell_cut((size_t)nranks + 1);
    cell_cut[0] = LLONG_MIN; cell_cut[nranks] = LLONG_MAX;
    if (cuts) {
        for (int r = 1; r < nranks; r++) cell_cut[r] = (long long)std::floor(cuts[r - 1] * inv_cell);
    } else {
        if (int r = fetch_records(c)) return r;
        std::vector<long long> cells((size_t)n);
        for (int i = 0; i < n; i++) {
            const PubRec &q = c->h_rec[i];
            cells[i] = (long long)std::floor((axis == 0 ? q.px : (axis == 1 ? q.py : q.pz)) * inv_cell);
        }
        std::sort(cells.begin(), cells.end());
        for (int r = 1; r < nranks; r++) cell_cut[r] = cells[(size_t)((long long)r * n / nranks)];
    }
    // an INTERIOR rank needs two layers of cells: with one, an agent that migrates into it lands in a cell that is the next rank's
    // halo layer as well, and that rank hears of it one step late (only the sender's two slab neighbours get messages)
    for (int r = 2; r < nranks; r++)
        if (cell_cut[r] - cell_cut[r - 1] < 2) { c->err = "partition: the cuts leave an interior rank fewer than two layers of cells (too many ranks for this swarm along this axis)"; return SCA_ERR_ARG; }
    if (nranks > 1 && cell_cut[1] == LLONG_MIN) { c->err = "partition: bad cut"; return SCA_ERR_ARG; }
    PartDev &P = c->part;
    for (int k = 0; k < 2; k++) CHK(c, hipMalloc((void **)&P.present[k], sizeof(int32_t) * n));
    CHK(c, hipMalloc((void **)&P.halo_tmp, sizeof(int32_t) * n));
    CHK(c, hipMalloc((void **)&P.counts, sizeof(int32_t) * 16));
    CHK(c, hipMalloc((void **)&P.emig, n));
    for (int k = 0; k < 2; k++) CHK(c, hipMemsetAsync(P.present[k], 0, sizeof(int32_t) * n, c->stream));   // entries beyond the exact counts are
    CHK(c, hipMemsetAsync(P.halo_tmp, 0, sizeof(int32_t) * n, c->stream));                                  // never meant to be read; if a bound
    CHK(c, hipMemsetAsync(P.emig, 0, n, c->stream));                                                        // slips, the read is agent 0, not garbage
    CHK(c, hipHostMalloc((void **)&c->part_host, sizeof(int) * 8));
    CHK(c, hipEventCreateWithFlags(&c->part_ev, hipEventDisableTiming));
    c->part_pending = false; c->part_age = 0;
    P.cur = 0; P.axis = axis; P.inv_cell = inv_cell;
    P.lo_cell = cell_cut[rank]; P.hi_cell = cell_cut[rank + 1];
    P.has_peer[0] = rank > 0; P.has_peer[1] = rank + 1 < nranks;
    P.cap_halo = cap_halo > 0 ? cap_halo : std::max(1024, n / 4);
    P.cap_mig = cap_mig > 0 ? cap_mig : std::max(256, n / 16);
    P.trk_st = c->trk_on ? c->trk.st : nullptr;
    P.trk_nbr0 = c->trk_on ? c->trk.nbr0 : nullptr;
    c->part_on = true; c->part_rank = rank; c->part_nranks = nranks;
    return part_classify(c);
}
int sca_partition_disable(sca_ctx *c) {
    API_ENTER(c);
    return part_free(c);
}
int64_t sca_partition_message_bytes(sca_ctx *c) {
    if (!c || !c->part_on) return 0;
    return (int64_t)part_message_bytes(c->part.cap_halo, c->part.cap_mig);
}
int sca_partition_counts(sca_ctx *c, int *owned, int *halo) {
    API_ENTER(c);
    if (!c->part_on) { c->err = "sca_partition_init first"; return SCA_ERR_STATE; }
    int h[5];
    CHK(c, hipMemcpyAsync(h, c->part.counts, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    if (owned) *owned = h[0];
    if (halo) *halo = h[1];
    return part_report(c, h[4]);
}
int sca_partition_owned(sca_ctx *c, int32_t *ids, int *count) {
    API_ENTER(c);
    ARG(c, ids && count);
    if (!c->part_on) { c->err = "sca_partition_init first"; return SCA_ERR_STATE; }
    CHK(c, hipMemcpyAsync(count, c->part.counts, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    CHK(c, hipMemcpyAsync(ids, c->part.present[c->part.cur], sizeof(int32_t) * (size_t)*count, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int sca_partition_pack(sca_ctx *c, void *device_buf_lower, void *device_buf_upper) {
    API_ENTER(c);
    if (!c->part_on) { c->err = "sca_partition_init first"; return SCA_ERR_STATE; }
    ARG(c, (device_buf_lower || !c->part.has_peer[0]) && (device_buf_upper || !c->part.has_peer[1]));
    c->part.trk_st = c->trk_on ? c->trk.st : nullptr;
    c->part.trk_nbr0 = c->trk_on ? c->trk.nbr0 : nullptr;
    const int cnt = std::max(1, c->d.shard_count);
    hipLaunchKernelGGL(k_part_pack, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, c->d, c->part,
                       c->part.has_peer[0] ? (uint8_t *)device_buf_lower : nullptr, c->part.has_peer[1] ? (uint8_t *)device_buf_upper : nullptr);
    CHK(c, hipGetLastError());
    return 0;
}
int sca_partition_unpack(sca_ctx *c, const void *device_buf_lower, const void *device_buf_upper) {
    API_ENTER(c);
    if (!c->part_on) { c->err = "sca_partition_init first"; return SCA_ERR_STATE; }
    ARG(c, (device_buf_lower || !c->part.has_peer[0]) && (device_buf_upper || !c->part.has_peer[1]));
    if (!c->part.has_peer[0] && !c->part.has_peer[1]) return 0;
    const int lanes = std::max(c->part.cap_halo, c->part.cap_mig);
    hipLaunchKernelGGL(k_part_unpack, dim3((lanes + 255) / 256), dim3(256), 0, c->stream, c->d, c->part,
                       c->part.has_peer[0] ? (const uint8_t *)device_buf_lower : nullptr, c->part.has_peer[1] ? (const uint8_t *)device_buf_upper : nullptr);
    CHK(c, hipGetLastError());
    return 0;
}
int sca_partition_commit(sca_ctx *c) {
    API_ENTER(c);
    if (!c->part_on) { c->err = "sca_partition_init first"; return SCA_ERR_STATE; }
    const int cnt = std::max(1, c->d.shard_count);
    hipLaunchKernelGGL(k_part_keep, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, c->d, c->part);
    hipLaunchKernelGGL(k_part_close<true>, dim3(16), dim3(256), 0, c->stream, c->d, c->part);
    CHK(c, hipGetLastError());
    return part_adopt(c, false);
}

int sca_step_begin(sca_ctx *c, int neighbor_mode) {
    API_ENTER(c);
    if (!c->state_set) { c->err = "sca_set_state first"; return SCA_ERR_STATE; }
    if (c->part_on && neighbor_mode != SCA_NBR_GRID) { c->err = "the cell-owner partition is a mode of SCA_NBR_GRID"; return SCA_ERR_UNSUPPORTED; }
    if (int r = launch_policy(c, neighbor_mode, false, true)) return r;
    return auto_join(c);                                                  // (the caller may read anything between the two halves of a step)
}
int sca_step_end(sca_ctx *c) {
    API_ENTER(c);
    return launch_collide_finish(c, false);
}

int sca_synchronize(sca_ctx *c) {
    API_ENTER(c);
    CHK(c, hipStreamSynchronize(c->stream));
    if (c->part_on) { int h[5]; CHK(c, hipMemcpy(h, c->part.counts, sizeof(h), hipMemcpyDeviceToHost)); if (int r = part_report(c, h[4])) return r; }
    if (c->perm_on_device) { if (int r = check_kd_overflow(c)) return r; }
    if (c->profiling && c->pool_used >= 4) {
        double a = 0, b = 0;
        const int steps = c->pool_used / 4;
        for (int s = 0; s < steps; s++) {
            float t0 = 0, t1 = 0;
            CHK(c, hipEventElapsedTime(&t0, c->pool[4 * s], c->pool[4 * s + 1]));          // K1, on the stream it ran on
            CHK(c, hipEventElapsedTime(&t1, c->pool[4 * s + 2], c->pool[4 * s + 3]));      // k_solve (+ k_lp)
            a += t0; b += t1;
        }
        c->ms_nbr = (float)(a / steps); c->ms_solve = (float)(b / steps);
        c->pool_used = 0;
    }
    if (c->profiling && c->pool_kd_used >= 2) {
        if (c->kd_stream) CHK(c, hipStreamSynchronize(c->kd_stream));
        if (c->trk_stream) CHK(c, hipStreamSynchronize(c->trk_stream));
        double a = 0;
        const int builds = c->pool_kd_used / 2;
        for (int s = 0; s < builds; s++) {
            float t = 0;
            CHK(c, hipEventElapsedTime(&t, c->pool_kd[2 * s], c->pool_kd[2 * s + 1]));
            a += t;
        }
        c->ms_kd_build = (float)(a / builds);
        c->pool_kd_used = 0;
    }
    if (c->profiling && c->pool_xch_used >= 2) {
        double a = 0;
        const int steps = c->pool_xch_used / 2;
        for (int s = 0; s < steps; s++) {
            float t = 0;
            CHK(c, hipEventElapsedTime(&t, c->pool_xch[2 * s], c->pool_xch[2 * s + 1]));
            a += t;
        }
        c->ms_exchange = (float)(a / steps);
        c->pool_xch_used = 0;
    }
    if (c->profiling && c->pool_trk_used >= 2) {
        if (c->trk_stream) CHK(c, hipStreamSynchronize(c->trk_stream));
        double a = 0;
        const int steps = c->pool_trk_used / 2;
        for (int s = 0; s < steps; s++) {
            float t = 0;
            CHK(c, hipEventElapsedTime(&t, c->pool_trk[2 * s], c->pool_trk[2 * s + 1]));
            a += t;
        }
        c->ms_replan = (float)(a / steps);
        c->pool_trk_used = 0;
    }
    return 0;
}

int sca_set_profiling(sca_ctx *c, int on) {
    API_ENTER(c);
    c->profiling = on != 0;
    c->pool_used = 0;
    c->pool_trk_used = 0;
    c->pool_xch_used = 0;
    c->pool_kd_used = 0;
    c->prof_tick = 0;
    return 0;
}

int sca_agent_steps(sca_ctx *c, int64_t *count, int reset) {
    API_ENTER(c);
    std::vector<unsigned long long> parts(256 * 16);
    CHK(c, hipMemcpyAsync(parts.data(), c->d.agent_steps, sizeof(unsigned long long) * parts.size(), hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    unsigned long long v = 0;
    for (unsigned long long x : parts) v += x;
    if (count) *count = (int64_t)v;
    if (reset) {
        CHK(c, hipMemsetAsync(c->d.agent_steps, 0, sizeof(unsigned long long) * parts.size(), c->stream));
        CHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

#ifdef SCA_KB_TIMING
int sca_debug_read_ps(sca_ctx *c, int *out, int count) {      // debug builds only: k_kd_block's per-phase ticks
    CHK(c, hipStreamSynchronize(c->stream));
    CHK(c, hipMemcpy(out, c->kd.ps, sizeof(int) * count, hipMemcpyDeviceToHost));
    return 0;
}
#endif

#ifdef SCA_TIMELINE
// debug builds only (SCA_BUILD_DEFS=-DSCA_TIMELINE): the device-side timeline of the next TL_RING steps (tools/device_timeline.py)
int sca_debug_timeline_enable(sca_ctx *c) {
    API_ENTER(c);
    const size_t cells = (size_t)TL_KERNELS * TL_RING;
    if (!c->d.tl) CHK(c, hipMalloc((void **)&c->d.tl, sizeof(unsigned long long) * 2 * cells));
    std::vector<unsigned long long> init(2 * cells);
    for (size_t i = 0; i < cells; i++) { init[2 * i] = ~0ull; init[2 * i + 1] = 0ull; }
    CHK(c, hipDeviceSynchronize());
    CHK(c, hipMemcpy(c->d.tl, init.data(), sizeof(unsigned long long) * 2 * cells, hipMemcpyHostToDevice));
    c->d.tl_step = -1;                                                  // the next pass is ring entry 0
    return 0;
}
int sca_debug_timeline_read(sca_ctx *c, unsigned long long *out /*[TL_KERNELS][TL_RING][2]*/, int *kernels, int *ring) {
    if (!c || !c->d.tl) return SCA_ERR_ARG;
    CHK(c, hipDeviceSynchronize());
    CHK(c, hipMemcpy(out, c->d.tl, sizeof(unsigned long long) * 2 * TL_KERNELS * TL_RING, hipMemcpyDeviceToHost));
    *kernels = TL_KERNELS; *ring = TL_RING;
    return 0;
}
#endif

#ifdef SCA_KT_TIMING
int sca_debug_read_kdq(sca_ctx *c, int *out, int count) {     // debug builds only: k_track's per-phase ticks
    CHK(c, hipDeviceSynchronize());
    CHK(c, hipMemcpy(out, c->d.kdq_list, sizeof(int) * (count < 16 ? count : 16), hipMemcpyDeviceToHost));
    if (count >= 48) CHK(c, hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(sca_dubins::g_td_ticks), sizeof(int) * 32));
    return 0;
}
#endif

int sca_selftest_l3norm(sca_ctx *c, int n, const double *a, const double *b, double *fast, double *exact) {
    API_ENTER(c);
    ARG(c, n > 0 && a && b && fast && exact);
    double *da = nullptr, *db = nullptr, *df = nullptr, *de = nullptr;
    CHK(c, hipMalloc((void **)&da, sizeof(double) * 3 * n)); CHK(c, hipMalloc((void **)&db, sizeof(double) * 3 * n));
    CHK(c, hipMalloc((void **)&df, sizeof(double) * n)); CHK(c, hipMalloc((void **)&de, sizeof(double) * n));
    CHK(c, hipMemcpyAsync(da, a, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
    CHK(c, hipMemcpyAsync(db, b, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_selftest_l3norm, dim3((n + 63) / 64), dim3(64), 0, c->stream, da, db, n, df, de);
    CHK(c, hipMemcpyAsync(fast, df, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipMemcpyAsync(exact, de, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(df); (void)hipFree(de);
    return 0;
}

// the restated glibc functions (sca_glibc_math.h): fn 0 sin(a), 1 cos(a), 2 acos(a), 3 atan2(a, b), 4 pow(a, 2) in the branch-free
// forms the kernels call; 5 sin, 6 cos, 7 atan2, 8 pow as the literal restatements (glibc's control flow); 9 / 10 the sine / cosine
// of the fused sincos; 11 atan2, 12 sin, 13 cos, 14 pow(a, 2) as the policy epilogue and the env update call them (sca_core.h m_*)
static double libm_eval_host(int fn, double a, double b) {
    double s2, c2;
    switch (fn) {
    case 0: return sca_gm::g_sin(a);
    case 1: return sca_gm::g_cos(a);
    case 2: return sca_gm::g_acos(a);
    case 3: return sca_gm::g_atan2(a, b);
    case 4: return sca_gm::g_pow2(a);
    case 5: return sca_gm::g_sin_ref(a);
    case 6: return sca_gm::g_cos_ref(a);
    case 7: return sca_gm::g_atan2_ref(a, b);
    case 8: return sca_gm::g_pow2_ref(a);
    case 9: sca_gm::g_sincos(a, s2, c2); return s2;
    case 10: sca_gm::g_sincos(a, s2, c2); return c2;
    case 11: return sca::m_atan2(a, b);
    case 12: sca::m_sincos(a, s2, c2); return s2;
    case 13: sca::m_sincos(a, s2, c2); return c2;
    default: return sca::m_pow2(a);
    }
}
int sca_selftest_libm_host(int fn, int n, const double *a, const double *b, double *out) {
    if (fn < 0 || fn > 14 || n < 0 || !a || !out || ((fn == 3 || fn == 7 || fn == 11) && !b)) return SCA_ERR_ARG;
    for (int i = 0; i < n; i++) out[i] = libm_eval_host(fn, a[i], b ? b[i] : 0.0);
    return 0;
}
int sca_selftest_libm(sca_ctx *c, int fn, int n, const double *a, const double *b, double *out) {
    API_ENTER(c);
    ARG(c, fn >= 0 && fn <= 14 && n > 0 && a && out && ((fn != 3 && fn != 7 && fn != 11) || b));
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    for (double **p : {&da, &db, &dout}) CHK(c, hipMalloc((void **)p, sizeof(double) * n));
    CHK(c, hipMemcpyAsync(da, a, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    if (b) CHK(c, hipMemcpyAsync(db, b, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    else CHK(c, hipMemsetAsync(db, 0, sizeof(double) * n, c->stream));
    hipLaunchKernelGGL(k_selftest_libm, dim3((n + 255) / 256), dim3(256), 0, c->stream, fn, da, db, n, dout);
    CHK(c, hipMemcpyAsync(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    for (double *p : {da, db, dout}) (void)hipFree(p);
    return 0;
}

// ---- trajectory log (Agent.history_info, agent.py:75-77,126-148) kept in HBM --------------------------------------
int sca_history_enable(sca_ctx *c, int capacity_rows) {
    if (!c || capacity_rows < 0) return SCA_ERR_ARG;
    if (int r_ = api_enter(c)) return r_;
    if (c->n <= 0) { c->err = "sca_set_agents first"; return SCA_ERR_STATE; }
    CHK(c, hipStreamSynchronize(c->stream));
    if (c->d.hist) { CHK(c, hipFree(c->d.hist)); c->d.hist = nullptr; }
    c->d.hist_cap = 0; c->d.hist_row = 0;
    if (capacity_rows == 0) return 0;
    CHK(c, hipMalloc((void **)&c->d.hist, sizeof(HistRow) * (size_t)capacity_rows * (size_t)c->n));
    c->d.hist_cap = capacity_rows;
    return 0;
}
int sca_history_rows(sca_ctx *c, int *rows_logged, int *rows_dropped) {
    API_ENTER(c);
    const int r = c->d.hist ? c->d.hist_row : 0;
    if (rows_logged) *rows_logged = std::min(r, c->d.hist_cap);
    if (rows_dropped) *rows_dropped = std::max(0, r - c->d.hist_cap);
    return 0;
}
int sca_get_history(sca_ctx *c, int first_row, int nrows, int agent_begin, int agent_count, double *pos, double *heading, float *vel) {
    API_ENTER(c);
    if (!c->d.hist) { c->err = "sca_history_enable first"; return SCA_ERR_STATE; }
    const int have = std::min(c->d.hist_row, c->d.hist_cap);
    if (first_row < 0 || nrows < 0 || first_row + nrows > have || agent_begin < 0 || agent_count < 0 || agent_begin + agent_count > c->n) {
        c->err = "history window out of range"; return SCA_ERR_ARG;
    }
    if (nrows == 0 || agent_count == 0) return 0;
    std::vector<HistRow> tmp((size_t)nrows * agent_count);
    CHK(c, hipMemcpy2DAsync(tmp.data(), sizeof(HistRow) * (size_t)agent_count,
                            c->d.hist + (size_t)first_row * c->n + agent_begin, sizeof(HistRow) * (size_t)c->n,
                            sizeof(HistRow) * (size_t)agent_count, (size_t)nrows, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < tmp.size(); i++) {
        const HistRow &h = tmp[i];
        if (pos) { pos[3 * i] = h.px; pos[3 * i + 1] = h.py; pos[3 * i + 2] = h.pz; }
        if (heading) { heading[3 * i] = h.a; heading[3 * i + 1] = h.b; heading[3 * i + 2] = h.g; }
        if (vel) { vel[3 * i] = h.vx; vel[3 * i + 1] = h.vy; vel[3 * i + 2] = h.vz; }
    }
    return 0;
}

int sca_get_actions(sca_ctx *c, float *action) {
    API_ENTER(c);
    ARG(c, action);
    std::vector<float> tmp((size_t)c->n * 8);
    CHK(c, hipMemcpyAsync(tmp.data(), c->d.action, sizeof(float) * 8 * c->n, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->n; i++) std::memcpy(action + 7 * (size_t)i, tmp.data() + 8 * (size_t)i, 7 * sizeof(float));
    return 0;
}

int sca_get_neighbors(sca_ctx *c, int32_t *nbr_n, int32_t *nbr_id, uint8_t *nbr_kind, double *nbr_dsq, uint8_t *nbr_valid) {
    API_ENTER(c);
    const int n = c->n;
    std::vector<int32_t> ids((size_t)n * K_MAX);
    CHK(c, hipMemcpyAsync(ids.data(), c->d.nbr_id, sizeof(int32_t) * K_MAX * n, hipMemcpyDeviceToHost, c->stream));
    if (nbr_n) CHK(c, hipMemcpyAsync(nbr_n, c->d.nbr_n, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream));
    if (nbr_dsq) CHK(c, hipMemcpyAsync(nbr_dsq, c->d.nbr_dsq, sizeof(double) * K_MAX * n, hipMemcpyDeviceToHost, c->stream));
    if (nbr_valid) CHK(c, hipMemcpyAsync(nbr_valid, c->d.nbr_valid, n, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < ids.size(); i++) {
        const int32_t v = ids[i];
        const bool ob = v >= 0 && (v & NBR_OBSTACLE_BIT);
        if (nbr_id) nbr_id[i] = v < 0 ? -1 : (v & ~NBR_OBSTACLE_BIT);
        if (nbr_kind) nbr_kind[i] = ob ? 1 : 0;
    }
    return 0;
}

int sca_get_nbr0(sca_ctx *c, double *dsq0) {
    API_ENTER(c);
    ARG(c, dsq0 && c->agents_set);
    double *tmp = c->kd.kx;      // build scratch, free between passes
    hipLaunchKernelGGL(k_nbr0, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->d, tmp);
    CHK(c, hipMemcpyAsync(dsq0, tmp, sizeof(double) * c->n, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int sca_get_diag(sca_ctx *c, int32_t *diag, int32_t *status, double *vpref_used) {
    API_ENTER(c);
    const int n = c->n;
    std::vector<int32_t> tmp((size_t)n * 8);
    CHK(c, hipMemcpyAsync(tmp.data(), c->d.diag, sizeof(int32_t) * 8 * n, hipMemcpyDeviceToHost, c->stream));
    if (status) CHK(c, hipMemcpyAsync(status, c->d.status, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream));
    if (vpref_used) CHK(c, hipMemcpyAsync(vpref_used, c->d.vpref_used, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, c->stream));
    CHK(c, hipStreamSynchronize(c->stream));
    if (diag) for (int i = 0; i < n; i++) std::memcpy(diag + 5 * (size_t)i, tmp.data() + 8 * (size_t)i, 5 * sizeof(int32_t));
    return 0;
}

int sca_set_shard(sca_ctx *c, int begin, int count) {
    API_ENTER(c);
    ARG(c, begin >= 0 && count >= 0 && begin + count <= c->n);
    // with a communicator the shard IS rank * n / nranks: the in-place ncclAllGather of sca_run_steps relies on it
    if (c->part_on) { c->err = "sca_set_shard with the cell-owner partition active (sca_partition_disable first)"; return SCA_ERR_STATE; }
    if (c->comm) { c->err = "sca_set_shard with an active communicator (the shard follows from rank / nranks; sca_comm_destroy first)"; return SCA_ERR_STATE; }
    c->d.shard_begin = begin; c->d.shard_count = count;
    c->kd_ahead = false; c->kdq_last = -1; c->auto_backoff = 0;           // another shard: the AUTO passes' counts described the old one
    return 0;
}
int sca_public_records(sca_ctx *c, int which, void **device_ptr, int64_t *bytes_per_agent) {
    API_ENTER(c);
    if (device_ptr) *device_ptr = which ? (void *)c->d.rec_new : (void *)c->d.rec;
    if (bytes_per_agent) *bytes_per_agent = (int64_t)sizeof(PubRec);
    return 0;
}
int sca_bind_public_records(sca_ctx *c, void *current, void *moved, int64_t bytes_each) {
    API_ENTER(c);
    ARG(c, (current == nullptr) == (moved == nullptr));
    if (!c->agents_set) { c->err = "sca_set_agents first"; return SCA_ERR_STATE; }
    const size_t live = sizeof(PubRec) * (size_t)c->n;                    // only the n live records move, never max_agents
    if (current) {
        ARG(c, bytes_each >= (int64_t)live);
        // carry the live records over into the caller's buffer
        CHK(c, hipMemcpyAsync(current, c->d.rec, live, hipMemcpyDeviceToDevice, c->stream));
        CHK(c, hipStreamSynchronize(c->stream));
        c->d.rec = (PubRec *)current; c->d.rec_new = (PubRec *)moved;
    } else {
        CHK(c, hipMemcpyAsync(c->rec_own, c->d.rec, live, hipMemcpyDeviceToDevice, c->stream));
        CHK(c, hipStreamSynchronize(c->stream));
        c->d.rec = c->rec_own; c->d.rec_new = c->rec_new_own;
    }
    return 0;
}
int sca_set_stream(sca_ctx *c, void *hip_stream) {
    API_ENTER(c);
    CHK(c, hipStreamSynchronize(c->stream));                             // work already enqueued finishes where it was
    c->stream = (hipStream_t)hip_stream;                                 // NULL is HIP's null stream (torch's default stream), taken literally
    return 0;
}
int sca_use_own_stream(sca_ctx *c) {
    API_ENTER(c);
    CHK(c, hipStreamSynchronize(c->stream));
    c->stream = c->stream_own;
    return 0;
}
int sca_last_kernel_ms(sca_ctx *c, float *neighbors_ms, float *solve_ms, float *update_ms) {
    API_ENTER(c);
    if (neighbors_ms) *neighbors_ms = c->ms_nbr;
    if (solve_ms) *solve_ms = c->ms_solve;
    if (update_ms) *update_ms = c->ms_update;
    return 0;
}

}  // extern "C"
