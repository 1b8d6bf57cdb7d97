// Shared host/device plain structs for the qrgpu kernels (gfx950).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace qrgpu {

#define QR_QH 96                  // hard cap on the MPC working-set size (rows of S^-1 held in LDS)
#define QR_MAX_TYPES 4
#define QR_WARM_STRIDE 80

// status bits (mirror include/qrgpu.h)
#define QRGPU_ST_MPC_MAXITER_D  0x1
#define QRGPU_ST_MPC_INFEAS_D   0x2
#define QRGPU_ST_MPC_OVERFLOW_D 0x4
#define QRGPU_ST_MPC_NOTSPD_D   0x8
#define QRGPU_ST_WBC_MAXITER_D  0x10
#define QRGPU_ST_WBC_INFEAS_D   0x20
#define QRGPU_ST_VMC_MAXITER_D  0x40
#define QRGPU_ST_VMC_INFEAS_D   0x80
#define QRGPU_ST_BAD_TYPE_D     0x01000000   // type id out of range or never set up (bit 24: above the flag byte and the 16-bit iteration count)

// XCD-aware robot index.  Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md: blocks b and b+8
// share an XCD) and each XCD has its own L2.  The SoA inputs put 32 consecutive robots in one 128-B line, so with
// rid = blockIdx every line was fetched (and partially written) by all 8 L2s: 8-10x the algorithmic HBM bytes
// (profiles/r01_rocprofv3_summary.md).  Mapping XCD x to the contiguous robot range [x*chunk, (x+1)*chunk) keeps
// a line inside one L2.  Speed only: any placement gives the same results.  Launch with grid = 8*ceil(n/8).
__device__ static inline int xcd_robot_index(int block, int n)
{
    const int chunk = (n + 7) >> 3;
    const int rid = (block & 7) * chunk + (block >> 3);
    return ((block >> 3) < chunk && rid < n) ? rid : -1;
}

// SetupProblem arguments (QI/controllers/mpc/qr_mpc_interface.h:157) + leg geometry for J^T f.
struct MpcType {
    float dt, mu, fmax, mass;
    float inertia[3];
    float weights[12];
    float alpha;
    float hip_l, upper_l, lower_l;
};

struct MpcLaunch {
    MpcType type[QR_MAX_TYPES];
    int n;
    int horizon;
    int lds_bytes;
    // Longest-first dispatch (DESIGN.md "tail"): slot j of an XCD's chunk solves robot order[j]; the kernel records what
    // the robot cost this time (clock64 ticks >> 12, saturated) for qr_lpt_order_kernel.  Either may be null.
    const int *order;
    int *cost;                  // bits 0-7: cost in units of 4096 cycles, saturated; bit 8: `big`; bits 16-31: the same in units of 256 cycles
    int cost_ema;               // the cost written is the mean of this solve's and the word's previous value (there is a history for this batch)
    // Rescue pass (DESIGN.md "working-set capacity"): a four-wave solve whose working set outgrows its 64 lanes / its LDS appends the
    // robot to rescue_list (rescue_count[parity] entries); the single-wave variant then re-solves exactly those robots with the
    // whole CU's LDS (rescue_mode = 1: workgroup b takes list entry b).  The main launch zeroes the other parity's counter.
    int *rescue_count;
    int *rescue_list;
    int rescue_parity;
    int rescue_mode;            // 0 main pass, 1 trailing list launch (rescue list + planning), 2 planned list launch
    // Planned list (DESIGN.md "tail"): robots that needed the list pass in the last call (or came within a few rows of the main pass's LDS
    // allotment, or belong to the big class nls >= big_nls) are solved by a list launch of their own, issued on a second stream AT THE
    // START of the next call, beside the main launch (which skips them), instead of after it.  pre_list / pre_count[parity] / skip[robot] are
    // written by the trailing list launch's planning workgroups from the `big` bit each solve leaves in cost[robot] (bit 8).
    int *pre_count;             // [0], [1] list lengths by call parity, [2] planning workgroups done (last one publishes the hint)
    int *pre_hint;              // the same two lengths in pinned host memory (the host decides from them whether to issue the planned launch), or null
    int *pre_list;
    int pre_list_next;          // where the planning writes the NEXT list, as an offset from pre_list (entries): 0 = over this one (the planned launch that reads it is through
                                // before the trailing launch starts), the lane's other half in an overlapped tick at h > 11 (there it is not: the trailing launch follows the main pass only)
    unsigned char *skip;
    int big_nls;                // 0 = no class rule
    int big_margin;             // a solve that ends within this many rows of what the main pass's LDS holds puts its robot on the planned list
    int planned_stride;         // planned launch of the h <= 16 whole-CU kernel: workgroup b takes entries b, b + grid, ... (a list longer than the CUs it may have)
    int big_cost, big_cost_stay; // (0 = off) smoothed cost, in units of 256 cycles, from which a main-pass solve puts its robot on the planned list,
                                // and from which a list solve keeps it there (QRGPU_H16_TWO: the long poles of a tick get a whole CU)
    int lds_main;               // the main pass's LDS allotment (bytes), for the `big` decision of a solve that runs in a list launch
    // the rescue launch also carries the longest-first sort of the next call (workgroups 0-7) when both are on: one launch fewer
    const int *lpt_cost_in;
    int *lpt_order_out;
    // h = 16: an all-stance inverse Hessian (150 KB) leaves LDS for 26 rows of S^-1; robots that need more keep S^-1 in this global
    // scratch instead ([robot][tri(QR_QH)] doubles, L2-resident), MAXB = 9 variants only
    double *sinv_spill;
    double warm_uthr;           // rows enter the next tick's guess only when their multiplier exceeds this fraction of the solve's largest (0 = all)
    int *started;               // planned list launch only: bumped by each of its workgroups as it starts (the gate in front of the main pass waits for them)
    int no_block_drop;          // diagnostic (QRGPU_NO_BLOCK_DROP=1): a warm start's wrong rows leave one downdate round at a time
    int no_wcache;              // diagnostic (QRGPU_NO_WCACHE=1): always take the z = w - M (N_A r) form
    // warm start (speed only): [robot][QR_WARM_STRIDE] bytes: the 6-bit active-row mask of each of the <= 64 original leg-steps at the end of the
    // slot's last solve, the contact-table bits it belonged to (8 bytes at offset 64), the horizon as a validity tag in the last byte; or null
    unsigned char *warm;
    double *flops;              // [robot][4] executed-arithmetic counts of the solve (qrgpu_mpc_flop_counts), or null
    int type_ready;             // bit t: type t was set up (robots naming any other type are flagged QRGPU_ST_BAD_TYPE)
    int epilogue;               // QRGPU_EPILOGUE_* bits applied to g_tau (MPC-only batches; 0 inside the fused tick)
    int hess_mode;              // K4 arithmetic: 0 = fp32 matrix instruction (exact fmaf chain), 1 = three-limb bf16 on the bf16 matrix instruction
    // Pipelined tick (qrgpu_tick_batch): the WBC launch runs BESIDE the MPC launches on a stream of its own and takes a robot's
    // forces as soon as that robot's solve has stored them.  done_flag[robot] = (tick epoch << 1) | on-the-rescue-list, written by the solve
    // with an agent-scope store behind its write-through (sc1) output stores; main_started: bumped by every workgroup of the main pass as it
    // starts (the WBC launch is gated on the whole main pass being resident, so that it can never take a CU from a solve it waits for).
    // Persistent main pass: the launch has one workgroup per resident slot and each takes robots off eight per-XCD queues (heads in qhead[8],
    // this launch's; qhead_next[8] are zeroed for the next one) until all are empty, its own XCD's first.  Hardware dispatch hands workgroup b
    // to XCD b % 8 in launch order and waits for a slot THERE: a free slot elsewhere stays empty meanwhile (6 us per second-round robot on
    // average, 40 us at worst: scratch/diag_slots.py), and a new workgroup takes 3 us to come up.
    int persist;
    int *qhead, *qhead_next;
    // The planned launch's own gate (no event from the context's stream: a one-thread launch on the side stream polls a "go" the context's
    // stream gives when it reaches this call) is bounded; should it give up, it leaves plan_epoch in *plan_abort: the planned workgroups then
    // leave at once and the main pass solves the robots it would have skipped.  Null: the launch was forked with an event.
    const int *plan_abort;
    int plan_epoch;
    // Pipelined tick: the trailing list launch does not wait for the planned launch through a stream event (8 us between the main pass and the
    // trailing launch even when the planned launch has long finished) but polls this count -- every workgroup of the planned launch bumps it
    // once, behind its written-through stores -- until planned_expect (cumulative, the host's running total).  Null: an event joins the streams.
    int *planned_done;
    int planned_expect;
    long long *tl;              // diagnostic (qrgpu_debug_timeline), or null
    int *ftime;                 // pipelined tick: when each robot's solve raised its flag (low word of the 100 MHz clock) -> the WBC launch's order next tick
    int *wbc_order_out;         // (trailing list launch / qr_lpt_order_kernel) that order, written for the next tick
    unsigned *done_flag;
    unsigned done_epoch;
    int *main_started;
    // Overlapped ticks (qrgpu_set_tick_overlap): tick t + 1's launches run on another stream set and start in the slots tick t's drain leaves empty.
    // What a robot carries from one solve to the next -- the warm-start words and the smoothed cost -- is handed over PER ROBOT: a solve stores
    // them written through (sc1), waits for the stores and leaves its tick's epoch in solved[robot]; the robot's next solve polls that word for
    // prev_epoch (bounded: 20 ms, then the robot starts cold and carries QRGPU_ST_PIPE_TIMEOUT -- the warm start is speed only) right before it
    // reads the words, with loads of the same kind.  cost_in: the buffer the previous tick's solves wrote (the smoothing reads it; this tick
    // writes `cost`, which the tick's trailing launch sorts while the NEXT tick's solves already write the other one).  All null / equal to
    // `cost`: every other launch.
    const int *cost_in;
    unsigned *solved;
    unsigned solved_epoch;
    const unsigned *prev_solved;
    unsigned prev_epoch;
    long long xtick_wait;       // bound of that wait, in ticks of the 100 MHz clock (20 ms; QRGPU_OV_WAIT_US for the give-up tests)
    // Overlapped ticks at h > 11: the trailing launch only sorts and plans -- eight small workgroups on the lane's stream
    // (plan_only = 1); a whole-CU workgroup would queue behind the NEXT tick's planned launch on the reserved CUs for a third of a tick.
    int plan_only;
    // ... and who solves a robot that turns up on the rescue list of such a tick -- one that changed class since the lane's plan was made, or whose
    // working set outgrew the main pass: the tick's PLANNED launch.  Its workgroups hold the reserved CUs anyway; when their share of the list is
    // done they stay and take rescue-list entries as the main pass appends them (rescue_taken: the list's second head, a compare-and-swap per
    // entry; an entry reads -1 until its writer's store has landed, and is set back to -1 by whoever takes it), until every workgroup of the main
    // pass has left (main_done, cumulative like `started`, against main_done_expect) and the list is empty.  Not the trailing launch on the
    // reserved CUs: the NEXT tick's planned workgroups hold those by then, each waiting for its robot's previous solve -- one of which would be
    // the robot the trailing launch cannot start to solve (measured: every listed robot 20 ms late, tick after tick).
    int *main_done;
    int main_done_expect;
    int *rescue_taken;          // [0..1] the rescue list's second head, [2..3] the planned list's head, by parity
    int linger;                 // planned launch: its first `linger` workgroups stay for the hand-overs, the others leave when the planned list is empty
};
#define QRGPU_ST_PIPE_TIMEOUT_D 0x02000000   // pipelined tick: the WBC gave up waiting for this robot's MPC forces (never seen; never silent)

// Force-balance QP parameters (qrgpu_vmc_desc): ComputeContactForce's arguments that do not change per tick.
struct VmcType {
    float mass;
    float inertia[9];          // robot->totalInertia, Eigen column-major
    float acc_weight[6];
    float reg_weight, friction, fmin_ratio, fmax_ratio;
    float hip_l, upper_l, lower_l;
};
struct VmcLaunch {
    VmcType type[QR_MAX_TYPES];
    int n;
    const float *ratio;      // [8][n] per-leg fMinRatio[4], fMaxRatio[4] of the world-frame overload, or null (the type's scalar ratios)
};

// Velocity-estimator parameters (qrgpu_estimator_desc)
struct EstimatorDesc {
    float hip_l, upper_l, lower_l;
    float hip_offset[12];
    float time_step, accelerometer_variance, sensor_variance;
    int window;
    float body_height;
};

// Foothold heuristic parameters (qrgpu_foothold_desc)
struct FootholdDesc {
    float hip_offset[12], default_hip_position[12];
    float hip_l, swing_kp[3], foot_clearance;
};

// Open-loop gait generator parameters (qrgpu_gait_desc)
struct GaitDesc {
    float stance_duration[4], duty_factor[4], initial_leg_phase[4];
    int initial_leg_state[4];
    float contact_detection_phase_threshold, wait_time;
    int advanced_trot;
};

// Velocity-mode swing action parameters (qrgpu_swing_velocity_desc)
struct SwingVelDesc { float hip_pos_com[12], stance_duration[4], swing_kp[3], desired_height; };

// Walk gait generator parameters after the constructor's bookkeeping (qrgpu_walk_gait_desc -> qrgpu_api.hip)
struct WalkDesc {
    float duty_factor[4], initial_leg_phase[4], full[4];
    int initial_leg_state[4], state_index0[4];
    float contact_detection_phase_threshold, true_swing_start_in_swing;
    int nq; int que[4]; float ratio[4], accum[5];
};

// Bytes of LDS in front of the block-packed inverse Hessian (must match the carve in qr_mpc_kernel.hip).
// The four-wave active set needs the exchange buffers xz[4][NV], xr[4][64]; the single-wave one (h > 11 by default, and the
// rescue pass) the staging arrays wl, yl, rl and the sAct / sPos tables.
__host__ __device__ static inline size_t mpc_lds_fixed_bytes(int h, bool multi)
{
    const size_t NV = 12 * (size_t)h, NL = 4 * (size_t)h;
    size_t b;
    if (multi) b = 8 * (NV + 4 * NV + 4 * 64 + NL);       // gl xz xr fmk
    else b = 8 * (3 * NV + QR_QH + NL);                                // gl wl yl rl fmk
    b += 4 * (36 + 36 + 28 + NV + NL + 13 * (size_t)h + (4 * h <= 44 ? 36 : 0));   // sT sU sSt sTraj sGait sV, sJ (h <= 11 only: at h = 16 an
                                                                       // all-stance inverse Hessian fills the CU's LDS to the last 100 bytes)
    b += 4 * (NL + QR_QH);                                             // sLs sAct
    b += 2 * (6 * NL + ((6 * NL) & 1));                                // sPos
    b += 4 * 16;                                                       // sMisc (+ the control block of the control/worker loop)
    return (b + 7) & ~(size_t)7;
}

// Per-robot epoch words of the overlapped tick (MpcLaunch::solved, WbcPipe::wbc_done).  Epochs run 1 .. 2^30 - 1 and wrap; a word only ever moves
// FORWARD (modulo 2^30): should a robot's wait for its predecessor give up and its solve finish before the late predecessor's, that predecessor's
// older epoch must not overwrite the newer one -- the robot's next solve would wait for a value that never comes back, and so would every one after it.
__device__ static inline bool qr_epoch_reached(unsigned word, unsigned want) { return ((word - want) & 0x3fffffffu) < 0x20000000u; }
__device__ static inline void qr_epoch_raise(unsigned *p, unsigned e)
{
    unsigned old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (!qr_epoch_reached(old, e)) {
        if (__hip_atomic_compare_exchange_strong(p, &old, e, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
}

// Pipelined tick, WBC side (qr_wbc_kernel): flag / epoch: the per-robot flags the MPC solves raise (MpcLaunch::done_flag); list / list_count: a
// second pass over the robots of a list (the MPC's rescue list) instead of the whole batch.  All null / zero: the plain launch.
struct WbcPipe {
    const unsigned *flag;
    unsigned epoch;
    const int *list;
    const int *list_count;
    const int *gate_abort;      // holds this tick's epoch when the gate in front of the WBC launch gave up waiting for the main pass (a caller with a long
                                //   queue of its own work in front of the tick): that launch must not run -- it would read its inputs before the caller's
                                //   stream has produced them and write its outputs before the solves write theirs -- so its workgroups leave at once and the
                                //   second pass, behind the MPC launches on the context's stream, computes the whole batch: the serial tick.  Or null
    int second;                 // this launch is the second pass (list / list_count may be null: nothing to do unless the gate gave up)
    int *finished;              // counted up once by either wave of a robot's workgroup when its outputs are in memory (written through): the tick's
                                //   join is a one-thread launch on the context's stream that waits for 2 n more of these (cumulative, never cleared), or null
    int *tlr;                   // diagnostic: [4][n] per robot, low word of the clock: WBC workgroup started, flag seen, done (last tick only), or null
    long long *tl;              // diagnostic (qrgpu_debug_timeline): [64 epochs][8] first / last moments of a tick's launches on the 100 MHz clock, or null
    const int *order;           // slot -> robot inside each XCD chunk: the robots in the order their solves ended in the last tick (MpcLaunch::ftime), or null
    // Overlapped ticks: the orientation task's memory (g_prev, quirk 4) is the one thing a robot's WBC carries from tick to tick, and the second
    // pass of tick t (its stream: the lane's) is not ordered against the first pass of tick t + 1 (the WBC stream).  So a workgroup leaves its
    // tick's epoch in wbc_done[robot] behind its written-through g_prev, and -- wait_epoch != 0 -- polls that word for the previous tick's epoch
    // (bounded, flagged) before it reads g_prev.  Null / 0: every other launch.
    unsigned *wbc_done;
    unsigned wait_epoch;
    long long wait_ticks;       // bound of that wait (100 MHz clock)
    // ... and there is NO second pass: a thousand (empty) workgroups dispatched one freed slot at a time on a machine that is never empty held the
    // tick's join back by 100 us.  The trailing MPC list launch -- half-CU workgroups in this mode, so a waiting WBC workgroup can never keep it
    // from starting -- raises the flag of a robot it has re-solved with bit 0 clear, and the robot's workgroup here waits on through the
    // "on the list pass" value for that (bounded by wait_ticks, flagged).
    int wait_list;
    long long flag_ticks;       // bound of the wait for the robot's forces (100 MHz clock; 4 ms, QRGPU_PIPE_WAIT_US for the tests)
    // Large batches, LABORATORY (QRGPU_WBC_CHUNKS; measured slower, LAB_NOTES A.7): the WBC launch is cut into launches of 1024 workgroups, each behind a gate of its own that
    // opens when the main pass has started that many workgroups more -- workgroup b of launch k stands for workgroup k * 1024 + b of ONE launch
    // (slot_base), through the main pass's own dispatch order (`order`: the longest-first order of THIS tick), so the robots of launch k are exactly
    // the robots whose solves the main pass has started by then.  One launch behind one gate starts when the main pass's LAST workgroup has: at 8192
    // robots that is 1.2 ms into a 1.65 ms tick, and 0.34 ms of WBC trail the last solve.
    int slot_base;
};

// WBC per-type constants (device buffer): BuildDynamicModel (QS/robots/qr_robot_a1_sim.cpp:176-343)
// reduced to rigid-body parameters, plus the controller gains (include/qrgpu.h qrgpu_model_desc).
// rb[k] = {m, h[3] (= m*com), Ibar[6] (about the link origin: xx yy zz xy xz yz)}.
#define QR_RB_BASE      0   // floating base link
#define QR_RB_BASE_EFF  1   // + the four abad rotors (constant in the base frame)
#define QR_RB_ABAD      2   // +side: 0 = right legs (mirrored), 1 = left legs
#define QR_RB_ABAD_EFF  4   // + hip rotor
#define QR_RB_HIP       6
#define QR_RB_HIP_EFF   8   // + knee rotor
#define QR_RB_KNEE      10  // same link inertia on both sides (:322 leaves it unmirrored)
struct WbcConst {
    double rb[11][10];
    double abad_loc[3];           // (0.1805, 0.047, 0) before leg signs
    double hip_l, upper_l, lower_l, foot_y;
    double k_rot;                 // rotor rotational inertia = Srot' Irot Srot (gear ratio 1)
    double hiprot_ex, hiprot_ey;  // E_rot^T e_y of the hip rotor frame (Rz(pi))
    double max_fz;                // totalNonRotorMass() * 9.81   (qr_single_contact.cpp:31)
    double kp_pos, kd_pos, kp_ori, kd_ori, kp_foot, kd_foot;
    double w_fb, w_fr, mu;
};

}  // namespace qrgpu
