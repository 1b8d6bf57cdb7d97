// Internal: the context behind the opaque qrgpu_ctx of include/qrgpu.h (shared by qrgpu_api.hip and qrgpu_comm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../include/qrgpu.h"
#include "qr_device_types.h"

using namespace qrgpu;

// One "lane" of MPC launch scheduling: everything a sequence of MPC launches on ONE stream carries from launch to launch -- the longest-first
// order, the rescue and planned lists with their ping-pong parities and pinned length hints, the cumulative counters its gates poll, the
// per-robot flags its solves raise for the WBC launch running beside them.  Lane 0 is the context's own stream (every call but an overlapped
// tick).  Lanes 1 and 2 have streams of their own and alternate between consecutive OVERLAPPED ticks (qrgpu_set_tick_overlap): tick t + 1's
// launches are queued on the other lane and start filling the slots tick t's drain leaves empty; what a robot carries from tick to tick
// (warm-start words, cost word, the orientation task's memory) is handed over per robot (MpcLaunch::solved, WbcPipe::wbc_done).
#define QR_LANES 5                 // 0: the context's stream; 1, 2: overlapped ticks at h <= 11; 3, 4: overlapped ticks at h > 11 (CU-masked streams)
#define QR_ABORT_RING 8            // give-up words are rings indexed by epoch: several ticks may be in flight behind a backlog
struct Lane {
    hipStream_t stream = nullptr;             // (lane 0: mirrors qrgpu_ctx::stream)
    bool own_stream = false;
    hipStream_t side_stream = nullptr;        // the planned list launch beside the main pass
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int *d_order = nullptr;                   // [max_batch] longest-first dispatch order of the next MPC launch of this lane
    int lpt_n = 0;                            // batch size d_order is valid for (0 = no history yet)
    int *d_rescue = nullptr;                  // [2] counters (ping-pong by call parity) + [max_batch] robot ids of the MPC rescue pass
    int rescue_parity = 0;
    bool last_rescue_active = false;          // did the last launch_mpc carry a trailing list launch (so that flags may say "on the rescue list")?
    int last_rescue_parity = 0;
    // planned list (launched beside the main pass on side_stream): [4] counters + [max_batch] robot ids; skip flags; batch size the plan is for
    int *d_pre = nullptr;
    unsigned char *d_skip = nullptr;
    int plan_n = 0;
    int *d_pre_hint = nullptr;                // device-side address of h_pre_count
    int *h_pre_count = nullptr;               // pinned: [0], [1] the planned list's length as of the last call, by parity (stale by a call or two at worst)
    int plan_sync_left = 0;                   // calls after a history reset that still end with a stream sync (so that the host sees the first plans' lengths)
    int *d_started = nullptr;                 // workgroups of planned list launches that have started, ever (qr_gate_kernel); never cleared
    unsigned started_total = 0;               // what that counter reaches once every planned launch issued so far has started (wraps like the counter)
    int *d_go = nullptr;                      // [0] "go" count of the planned launches' gates (cumulative); [1 + (plan epoch & 7)]: plan epoch of a gate that gave up
    unsigned go_total = 0;
    int plan_epoch = 0;
    int *d_planned_done = nullptr;            // workgroups of planned launches that are through, ever (polled by the trailing launch of a pipelined tick)
    unsigned planned_done_total = 0;
    int *d_qhead = nullptr;                   // [2][8] queue heads of the persistent main pass, ping-pong (a launch zeroes the other half)
    int qhead_parity = 0;
    int two_hold = 0;                         // h > 11 two to a CU: calls left on one workgroup per CU after the planned list outgrew 45 % of the batch
    bool two_probe = false;                   //   ... and the call after them runs two to a CU whatever the count says, to get a fresh plan
    unsigned *d_done_flag = nullptr;          // [max_batch] (tick epoch << 1) | on-the-rescue-list, raised by this lane's solves for the WBC launch
    float *d_cmd_tick = nullptr;              // [12][max_batch] force scratch of a tick whose caller passes no force array
    bool masked = false;                      // lanes 3, 4: stream = all CUs but the reserved ones, side_stream = the reserved ones (both owned)
    int *d_lane_done = nullptr;               // overlapped ticks of this lane whose tail (second WBC pass) is through, ever (the tick's join polls it)
    int *d_main_done = nullptr; int main_done_total = 0;   // h > 11 overlapped: workgroups of the lane's main passes that have left (cumulative), MpcLaunch::main_done
    int order_parity = 0;                               // which half of d_order the lane's next main pass reads (the trailing launch writes the other)
    const int *order_used = nullptr;                    // ... and what the last main pass read (the chunked WBC launches of that tick read it too)
    bool join_recorded = false;                         // ev_join has been recorded at least once (the lane's next overlapped tick at h > 11 waits for it)
    int last_linger = 0;                                // how many workgroups of the lane's last planned launch stay until its main pass is through
    int *d_rescue_taken = nullptr;                      // ... and the rescue list's second head, per parity (MpcLaunch::rescue_taken)
    unsigned lane_done_total = 0;
};

struct qrgpu_ctx {
    int device = 0;
    int max_batch = 0;
    int horizon_max = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    MpcLaunch mpc{};
    bool mpc_ready[QR_MAX_TYPES] = {false, false, false, false};
    bool wbc_ready[QR_MAX_TYPES] = {false, false, false, false};
    VmcLaunch vmc{};
    bool vmc_ready[QR_MAX_TYPES] = {false, false, false, false};
    WbcConst wbc_host[QR_MAX_TYPES];
    WbcConst *d_wbc = nullptr;
    bool wbc_dirty = true;
    // scratch for the single-robot calls and the fused tick
    // Staging of the single-robot calls.  Default: ONE block of pinned, mapped host memory that the kernels read and write in place (zero
    // copy: d_* are the device-side addresses of h_*), so a call is "fill h_in1, one launch, wait, read h_out1" with no copy command on the
    // stream.  QRGPU_SINGLE_COPIES=1: device buffers and three hipMemcpyAsync per call (round 2's form, kept for the A/B of INTEGRATION.md).
    float *d_in1 = nullptr;       // staging: single-robot inputs
    float *d_out1 = nullptr;      // staging: single-robot outputs
    int *d_st1 = nullptr;
    float *h_in1 = nullptr, *h_out1 = nullptr;     // host-side addresses of the same memory (zero copy) or null
    int *h_st1 = nullptr;
    void *h_stage = nullptr;      // the pinned block
    bool zero_copy = false;
    int type_stage = 0;           // (copy path) the type word in flight
    Lane lane[QR_LANES];
    // what each robot cost in the last MPC launch: [0] every launch of lane 0, in place; an overlapped tick of epoch e writes [e & 1] and smooths
    // with [(e & 1) ^ 1], its predecessor's (the trailing launch of tick t sorts a buffer that tick t + 1's solves do not write)
    int *d_cost[2] = {nullptr, nullptr};
    int cost_n[2] = {0, 0};       // batch size each holds the costs of a whole overlapped tick for (0: not)
    double *d_sinv_spill = nullptr;   // [max_batch][tri(QR_QH)] S^-1 scratch of the h > 11 variants, allocated at first use
    bool planned = true;
    int big_nls = 0;
    // pipelined tick (qrgpu_set_tick_pipeline, default on): the WBC launch of a tick runs on wbc_stream beside that tick's MPC launches
    bool pipeline = true;
    hipStream_t wbc_stream = nullptr;
    hipEvent_t ev_wbc_fork = nullptr, ev_wbc_join = nullptr;
    int *d_main_started = nullptr;            // main-pass workgroups started, ever (the WBC launch's gate); never cleared
    unsigned main_started_total = 0;
    unsigned tick_epoch = 0;
    int main_slots[16][2] = {};               // resident workgroups per CU of each main-pass variant at the LDS size it was last configured for (0: not asked yet)
    int main_slots_lds[16][2] = {};
    int *d_tick_done = nullptr;               // pipelined ticks complete (bumped by their joins), ever: what qrgpu_allgather_tau_of_tick's gate polls
    unsigned tick_done_total = 0;
    bool last_tick_piped = false;             // the context's most recent qrgpu_tick_batch was a pipelined one with a polling join
    int *d_gather_done = nullptr;             // [2] gathers finished per source-buffer slot, ever (qrgpu_allgather_fence polls it)
    unsigned gather_total[2] = {0, 0};
    unsigned gather_joined[2] = {0, 0};            // ... of which a pipelined tick's join has already made the compute stream wait for
    int *d_gate_abort = nullptr;              // [QR_ABORT_RING] word (epoch & 7): epoch of the pipelined tick whose WBC gate timed out (0: none)
    int *d_wbc_finished = nullptr;            // waves of pipelined WBC launches whose outputs are in memory, ever (the tick's join); never cleared
    unsigned wbc_finished_total = 0;
    // overlapped ticks (qrgpu_set_tick_overlap): consecutive pipelined ticks alternate between lanes 1 and 2
    int overlap = 0;                          // 0 off, 1 on (the caller's promise about inputs and output buffers: include/qrgpu.h)
    int ov_next = 0;                          // which of lanes 1 / 2 the next overlapped tick takes
    bool ov_chain = false;                    // the context's last launch was an overlapped tick (the next one may start under it)
    int ov_n = 0;                             // ... of this batch size
    unsigned ov_epoch = 0, ov_main_total = 0; // ... with this epoch, after which d_main_started reaches this
    const void *ov_out[4] = {nullptr, nullptr, nullptr, nullptr};   // ... writing these output arrays (force, tau, qdes, status)
    const void *ov_prev_ori = nullptr;        // ... and this orientation-task memory
    int ov_lane_last = 0;                     // ... on this lane
    int ov_fence_slots = 0;                   // bit s: qrgpu_allgather_fence(s) was called since the last overlapped tick (that tick's lane waits for the gather too)
    int ov_hold = 0;                          // calls left on the plain pipelined tick after a lane found a plan (a population with whole-CU robots)
    bool ov_prev_plan = false;                // ... with a planned launch (its successor is not chained either)
    // the tail of an overlapped tick -- its second WBC pass and the count its join polls -- runs on a stream of its own behind an event of the lane's
    // stream: on the lane's stream those thousand (empty) workgroups, dispatched one freed slot at a time on a machine that is never empty, stood
    // between the lane's next tick and its gate (60 us per tick)
    // the WBC launches of overlapped ticks: a stream of the highest priority.  Tick k's WBC workgroups and tick k + 1's solves want the same freed
    // slots; at equal priority the solves get most of them, tick k's WBC launch lasts until tick k + 1's main pass is dispatched (tick duration: two
    // periods) and tick k + 2, which waits for tick k's join, starts late every other tick
    hipStream_t wbc_stream_hi = nullptr;
    // overlapped ticks at h > 11: a tenth of a mixed shard wants a whole CU per robot, which a machine that is never empty does not offer -- so the
    // machine is split in space: ov16_side_cus CUs (a multiple of 32 on 256: every XCD's share a multiple of four, or the LDS a queue may use per CU
    // drops to ~100 KB) are reserved for the whole-CU launches (planned list, trailing launch), the main passes run on the others
    hipStream_t wbc_stream_16 = nullptr;      // ... and the WBC launches too (a WBC workgroup waiting on a reserved CU for a planned robot's forces would keep
                                              //     the whole-CU workgroup that computes them from ever starting there)
    int ov16_side_cus = 0;
    uint32_t mask16_main[16] = {}, mask16_side[16] = {};
    // a chained tick waits for what the caller had queued on the context's stream when it made the PREVIOUS tick call (its predecessor's predecessor's
    // join and whatever consumed that tick's outputs: the arrays this tick overwrites), recorded at that call: ring of two
    hipEvent_t ev_call[2] = {nullptr, nullptr};
    int ev_call_last = -1;                    // index of the event recorded at the last overlapped tick call (-1: none)
    long long *d_join_dbg = nullptr;          // diagnostic (qrgpu_debug_counters): what the last joins saw
    int ov_stats[2] = {0, 0};                 // overlapped ticks issued so far: chained to their predecessor / behind an event of the context's stream
    unsigned *d_solved = nullptr;             // [max_batch] epoch of the overlapped tick whose solve of the robot has left its warm-start and cost words in memory
    unsigned *d_wbc_done = nullptr;           // [max_batch] ... whose WBC pass has left the orientation task's memory (prev_ori) in memory
    int *d_tlr = nullptr;                     // diagnostic: [4][max_batch] per-robot WBC moments of the last pipelined tick
    long long *d_timeline = nullptr;          // diagnostic (qrgpu_debug_timeline): [64][8], or null
    int *d_ftime = nullptr;                   // [max_batch] when each robot's solve ended in the last pipelined tick (100 MHz clock, low word)
    int *d_wbc_order = nullptr;               // [2][max_batch] the WBC launch's slot -> robot map from those times, ping-pong: a tick's WBC launch reads one
    int wbc_order_parity = 0;                 //   half while the launch behind its main pass writes the other
    int wbc_order_n = 0;                      // batch size the half to be read next was written for (0: none)
    bool lpt = true;
    bool rescue = true;
    int epilogue = 0;             // QRGPU_EPILOGUE_* bits
    double *d_flops = nullptr;    // [max_batch][4], allocated by qrgpu_enable_flop_count
    bool flops_on = false;
    int flops_n = 0;              // robots of the last counted launch
    bool warm = true;             // warm start of the MPC active set from the slot's previous solve
    unsigned char *d_warm = nullptr;   // [max_batch][QR_WARM_STRIDE]
    int warm_n = 0;               // batch size d_warm is valid for (0 = nothing yet)
    bool last_main_persist = false;  // the last main pass took robots off queues (its "started" count is per robot, in no fixed order: no chunked WBC launches)
    void *d_dbg_cycles_wbc = nullptr;
    void *d_dbg_cycles = nullptr; // optional [max_batch][8] int64 phase stamps of the MPC kernel (qrgpu_debug_cycles)
    int lds_per_cu = 0, num_cu = 0;
    std::string name;
    std::string err;
    // multi-GPU: context-owned RCCL communicator and the stream its all-gathers run on (qrgpu_comm.hip)
    void *comm = nullptr;         // ncclComm_t
    bool comm_owned = false;
    int comm_nranks = 0, comm_rank = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_tick = nullptr;          // compute stream -> comm stream: this tick's torques are complete
    hipEvent_t ev_gather[2] = {nullptr, nullptr};   // comm stream -> compute stream: the gather that read buffer `slot` has landed
    bool ev_gather_pending[2] = {false, false};
    // timing
    bool timing = false;
    bool timing_paused = false;
    int timing_every = 1;          // events bracket every timing_every-th launch of a kernel (qrgpu_enable_timing(ctx, N))
    unsigned ev_calls[2] = {0, 0};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[2];
    size_t ev_used[2] = {0, 0};
    std::vector<hipEvent_t> marks;   // qrgpu_mark: caller-indexed events on the context stream
};

// Environment switches.  The SUPPORTED ones are listed in include/qrgpu.h (name, default, effect) and read with getenv.  Everything else that
// rounds 1-3 measured with -- alternative launch shapes, event / polling forms of each hand-over, thresholds -- is a LABORATORY switch: read
// through lab_env, which answers only when QRGPU_LAB=1 is set, so that a product run cannot be steered by a leftover of an experiment.
// qrgpu_create warns once per process about any other QRGPU_* variable it finds in the environment.
inline const char *lab_env(const char *name)
{
    static const bool lab = [] { const char *e = getenv("QRGPU_LAB"); return e && atoi(e) != 0; }();
    return lab ? getenv(name) : nullptr;
}
#define QRGPU_SUPPORTED_ENV "QRGPU_TICK_PIPELINE", "QRGPU_PIPE_GATE_MS", "QRGPU_PLAN_GO_MS", "QRGPU_PIPE_WAIT_US", "QRGPU_OV_WAIT_US", "QRGPU_OV_FAULT", "QRGPU_OV_PLAN_HOLD", "QRGPU_OV16", \
                            "QRGPU_COMM_EVENTS", "QRGPU_SINGLE_COPIES", "QRGPU_PERSIST", "QRGPU_H16_TWO", "QRGPU_H16_TWO_HOLD", "QRGPU_H16_BIG_US", "QRGPU_H16_BIG_STAY_US", \
                            "QRGPU_LIB", "QRGPU_EXTRA_FLAGS", "QRGPU_LAB"
#define QRGPU_LAB_ENV "QRGPU_WBC_ORDER", "QRGPU_WARM_UTHR", "QRGPU_TINY_WHOLE_CU", "QRGPU_SIDE_PRIORITY", "QRGPU_PLAN_SYNC", "QRGPU_PLANNED_WAVES", "QRGPU_PLANNED_MODE", "QRGPU_PLANNED_JOIN", "QRGPU_PLANNED_GATE", "QRGPU_PLANNED_FORK", "QRGPU_PLANNED_EXTRA", "QRGPU_PIPE_JOIN", "QRGPU_PIPE_FORK", "QRGPU_PIPE_EARLY", "QRGPU_OWN_STREAM", "QRGPU_OV_WBC_PRIORITY", "QRGPU_NO_WCACHE", "QRGPU_NO_BLOCK_DROP", "QRGPU_MAIN_WGS", "QRGPU_MAIN_THREADS", "QRGPU_H16_TWO_WAVES", "QRGPU_H16_THREADS", "QRGPU_COST_EMA", "QRGPU_BIG_MARGIN", "QRGPU_OV16_SIDE_CUS", "QRGPU_OV16_DEBUG", "QRGPU_OV16_LINGER", "QRGPU_OV16_WBC_MASK", "QRGPU_OV16_COST", "QRGPU_WBC_CHUNKS"

#define HIPCHK(ctx, call)                                                                    \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                  \
            return QRGPU_ERR_LAUNCH;                                                         \
        }                                                                                    \
    } while (0)

