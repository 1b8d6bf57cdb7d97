/* ============================================================================
 * qrgpu.h -- C ABI of libqrgpu.so: batched convex-MPC + WBC control ticks for
 * quadrupeds on AMD MI355X (gfx950).  Plain pointers and sizes only; no C++,
 * torch or HIP types appear in any signature (the stream is passed as void*).
 *
 * Each entry point replaces one seam of TopHillRobotics/quadruped-robot
 * ("QI/" = quadruped/include/quadruped/, "QS/" = quadruped/src/):
 *
 *   qrgpu_mpc_setup        <- Quadruped::SetupProblem          QI/controllers/mpc/qr_mpc_interface.h:157
 *   qrgpu_mpc_solve1       <- Quadruped::SolveMPCKernel + 12x GetMPCSolution   :200, :215
 *   qrgpu_mpc_solve_batch  <- the same, for n independent robots (MPCStanceLegController::SolveDenseMPC
 *                             + GetAction torque map, QS/controllers/mpc/qr_mpc_stance_leg_controller.cpp:385-410,129-156)
 *   qrgpu_wbc_setup        <- qrRobot*::BuildDynamicModel + qrWbcLocomotionController ctor gains
 *                             QS/robots/qr_robot_a1_sim.cpp:176-343, QS/controllers/wbc/qr_wbc_locomotion_controller.cpp:29-77
 *   qrgpu_wbc_run1         <- qrWbcLocomotionController<float>::Run   QI/controllers/wbc/qr_wbc_locomotion_controller.hpp:59
 *   qrgpu_wbc_run_batch    <- the same, for n robots
 *   qrgpu_mpc_frontend_batch <- MPCStanceLegController::SetupCommand/Run/UpdateMPC (reference trajectory + contact table)
 *                             QS/controllers/mpc/qr_mpc_stance_leg_controller.cpp:158-382
 *   qrgpu_vmc_setup / qrgpu_vmc_force_batch / qrgpu_vmc_force_world_batch <- Quadruped::ComputeContactForce (control-frame and world-frame
 *                             overloads, :190-301, :304-398) + qrRobot::MapContactForceToJointTorques
 *                             QS/controllers/balance_controller/qr_qp_torque_optimizer.cpp:190-301, QS/robots/qr_robot.cpp:241-251
 *                             (what TorqueStanceLegController::GetAction calls, qr_torque_stance_leg_controller.cpp:500-507)
 *   qrgpu_estimator_update_batch <- qrRobot::UpdateDataFlow (leg kinematics) + qrRobotVelocityEstimator::Update + qrRobotPoseEstimator::Update
 *                             QS/robots/qr_robot.cpp:62-72,187-197, QS/estimators/qr_robot_velocity_estimator.cpp:77-133,
 *                             QS/estimators/qr_robot_pose_estimator.cpp:68-165 (qrRobotEstimator::Update, qr_robot_estimator.cpp:79-83)
 *   qrgpu_gait_update_batch <- qrOpenLoopGaitGenerator::Update / Schedule   QS/gait/qr_openloop_gait_generator.cpp:126-249
 *   qrgpu_swing_targets_batch <- qrRaibertSwingLegController::GetAction (ADVANCED_TROT)   QS/controllers/qr_swing_leg_controller.cpp:362-424
 *   qrgpu_footholds_batch  <- qrRaibertSwingLegController::Update + qrFootholdPlanner::ComputeHeuristicFootHold
 *                             QS/controllers/qr_swing_leg_controller.cpp:211-236, QS/planner/qr_foothold_planner.cpp:110-239
 *   qrgpu_tick_batch       <- one MPC solve + one WBC tick per robot, WBC fed with that MPC's Fr_des
 *                             (QS/fsm/qr_fsm_state_locomotion.cpp:130-158 without the MPC/WBC time-slicing)
 *   qrgpu_set_torque_epilogue <- the motor-command tail of that state: abad compensation (:141-151) and the +-23 N m clip
 *                             (QS/fsm/qr_safety_checker.cpp:48-66)
 *   qrgpu_comm_* / qrgpu_allgather_tau <- no reference equivalent (the reference runs one robot per process): the all-gather of torques
 *                             of BASELINE.json's sharded configurations
 *
 * Error behaviour mirrors the reference: no exceptions; the reference prints
 * "failed to solve!" and carries on (qr_mpc_interface.cpp:440-442) -- here every
 * robot gets a status word instead and every call returns a qrgpu_error.
 *
 * All computation runs in hand-written HIP kernels.  There is NO CPU fallback:
 * without a usable gfx950 device qrgpu_create fails with QRGPU_ERR_NO_DEVICE.
 *
 * Batched layouts are structure-of-arrays, [field][robot] with the robot index
 * fastest (array element (f, i) at f*n + i), float32, device memory:
 *   mpc_state [28][n] : p[3], v_world[3], quat_wxyz[4], w_world[3], r[12] (3x4 column-major:
 *                       r[3*leg+axis], foot - CoM in the world-aligned frame), rpy[3]
 *   traj      [12h][n]: desired state per horizon step (rpy, xyz, omega, v), step-major
 *   gait      [4h][n] : contact table, step-major (row-major h x 4, as mpcTable)
 *   fb_state  [37][n] : quat_wxyz[4], pos[3], omega_body[3], v_body[3], q[12], qd[12]   (FBModelState)
 *   wbc_cmd   [67][n] : pBody_des[3], vBody_des[3], aBody_des[3], pBody_RPY_des[3], vBody_Ori_des[3],
 *                       pFoot_des[12], vFoot_des[12], aFoot_des[12], Fr_des[12], contact[4] (0/1)  (qrWbcCtrlData)
 *   prev_ori  [3][n]  : in/out, desiredVel of the body-orientation task from the previous WBC call
 *                       (stateful quirk of task_set/qr_task_body_orientation.cpp:68 vs :73)
 *   force     [12][n] : MPC ground-reaction forces of horizon step 0, world frame (= wbcData.Fr_des)
 *   tau       [12][n] : joint torques
 *   vmc_in    [37][n] : force-balance QP inputs: footPositionsInBaseFrame[12] (3*leg+axis), desiredAcc[6], contacts[4],
 *                       Rcb[9] (row-major; identity on PLANE / PLUM_PILES terrain), g.head(3) ((0,0,9.8) on a plane), surfaceNormal[3]
 *   est_in    [54][n] : estimator inputs: baseAccInBaseFrame[3], baseLinearAcceleration[3], quat_wxyz[4], rpyRate[3], footContact[4],
 *                       motor angles q[12], motor velocities dq[12], desiredLegState[4] (LegState values), groundOrientationMat[9]
 *                       (GetAlignedDirections, row-major; identity on a plane)
 *   est_out   [42][n] : filtered baseLinearAcceleration[3], baseVInWorldFrame[3], baseVelocityInBaseFrame[3], baseWInWorldFrame[3],
 *                       footPositionsInBaseFrame[12], footVelocitiesInBaseFrame[12] (3*leg+axis), basePosition[3] (odometry x, y and
 *                       the stance-foot height), heightInControlFrame (NaN = unchanged: no stance foot), absoluteHight, odometry yaw
 *   fe_in     [64][n] : front-end inputs per control tick: des_height, des_roll, des_pitch, x_vel_cmd, y_vel_cmd, yaw_vel_cmd
 *                       (stateDes 2,3,4,6,7,11 after UpdateDesCommand), basePosition[3], yaw, quat_wxyz[4], footPosWorld[12]
 *                       (leg-major), footTargetPositionsInWorldFrame[12], contacts[4], phaseInFullCycle[4], dutyFactor[4],
 *                       normalizedPhase[4], desiredLegState[4], legState[4] (LegState enum values as floats),
 *                       firstSwingBaseState x, y
 *   fe_state  [8][n]  : in/out controller memory: xVelDes, yVelDes, yawTurnRate, yawDesTrue, posDesiredinWorld[3], iterationCounter
 *
 * ENVIRONMENT SWITCHES.  These are the supported ones (read once per process).  qrgpu_create warns once about any other QRGPU_* variable it
 * finds: the alternative launch shapes and thresholds rounds 1-3 measured with are laboratory switches, ignored unless QRGPU_LAB=1 is set
 * (LAB_NOTES.md lists them with what each measured).
 *   name                    default  effect
 *   QRGPU_TICK_PIPELINE     1        0: qrgpu_tick_batch queues the WBC launch behind the MPC launches (the serial tick) whatever qrgpu_set_tick_pipeline says
 *   QRGPU_PIPE_GATE_MS      50       bound of the gate in front of a pipelined tick's WBC launch; one that gives up turns the tick into the serial one
 *   QRGPU_PLAN_GO_MS        50       bound of the planned launch's wait for its "go"; one that gives up calls the plan off for the call
 *   QRGPU_PIPE_WAIT_US      4000     bound of a WBC workgroup's wait for its robot's forces (then QRGPU_ST_PIPE_TIMEOUT)
 *   QRGPU_OV_WAIT_US        20000    overlapped ticks: bound of a robot's waits for its previous solve / WBC pass (then QRGPU_ST_PIPE_TIMEOUT)
 *   QRGPU_OV_PLAN_HOLD      31       overlapped ticks: calls on the plain pipelined tick after a lane found a planned list (0: never go back)
 *   QRGPU_OV16              1        overlapped ticks: 0 keeps contexts with a horizon beyond 11 on the plain pipelined tick
 *   QRGPU_OV_FAULT          0        test hooks of the give-up paths (tests/test_gpu_overlap.py): 1 makes every chained tick wait for an epoch nobody writes;
 *                                    2 (h > 11) sends the planned launch's workgroups home without the robots the main pass hands on
 *   QRGPU_COMM_EVENTS       unset    the all-gather's hand-overs: unset = stream events when the communicator has more than one rank, polled counts
 *                                    with one; 1 = events always; 0 = polled counts always
 *   QRGPU_SINGLE_COPIES     0        1: the single-robot calls stage through device buffers and three copies instead of one mapped pinned block
 *   QRGPU_PERSIST           1        h > 11: persistent main pass (one workgroup per resident slot taking robots off per-XCD queues); 0 off, 2 also at h <= 11
 *   QRGPU_H16_TWO           1        h > 11: two workgroups per CU from 3.5 robots per CU on; 0 never, 2 from 64 robots on
 *   QRGPU_H16_TWO_HOLD      31       ... calls on one workgroup per CU after the planned list outgrew 45 % of the batch
 *   QRGPU_H16_BIG_US        450      ... smoothed solve time from which a robot is planned onto a whole CU
 *   QRGPU_H16_BIG_STAY_US   300      ... and below which it leaves it again
 *   QRGPU_LIB               unset    (Python mirror, bench.py) path of another build of libqrgpu.so to load: A/B runs
 *   QRGPU_EXTRA_FLAGS       unset    (build.py) extra hipcc flags; -DQR_TIMELINE builds the per-tick stamps of qrgpu_debug_timeline in
 *   QRGPU_LAB               0        1: the laboratory switches are read
 *   GPU_MAX_HW_QUEUES       (HIP)    the HIP runtime's: hardware queues per process (its default 4; the Python mirror sets 8 when unset).  Streams that
 *                                    share a queue serialise each other: qrgpu_set_tick_overlap probes for it and refuses the mode
 * ========================================================================== */
#ifndef QRGPU_H
#define QRGPU_H

#ifdef __cplusplus
extern "C" {
#endif

#define QRGPU_MAX_HORIZON 16      /* K_MAX_GAIT_SEGMENTS, QI/controllers/mpc/qr_mpc_interface.h:33 */
#define QRGPU_MAX_TYPES   4       /* robot types (parameter sets) per context */

typedef struct qrgpu_ctx qrgpu_ctx;

typedef enum {
    QRGPU_OK = 0,
    QRGPU_ERR_NO_DEVICE = 1,      /* no gfx950 device / HIP runtime failure at create */
    QRGPU_ERR_BAD_ARG = 2,
    QRGPU_ERR_NOT_SETUP = 3,      /* solve before setup */
    QRGPU_ERR_LAUNCH = 4,         /* kernel launch or runtime error (see qrgpu_last_error) */
    QRGPU_ERR_ALLOC = 5,
    QRGPU_ERR_COMM = 6            /* RCCL failure (librccl missing, communicator error; see qrgpu_last_error) */
} qrgpu_error;

/* Per-robot status word written by the kernels: bits 0-7 and 24-31 are the flags below (0 = converged), bits 8-23
 * hold the number of MPC active-set iterations (diagnostic).  Test `(status & QRGPU_ST_FLAG_MASK) == 0`. */
#define QRGPU_ST_FLAG_MASK     0xff0000ffu
#define QRGPU_ST_ITERATIONS(s) (((s) >> 8) & 0xffff)
#define QRGPU_ST_OK            0
#define QRGPU_ST_MPC_MAXITER   0x1    /* active-set iteration cap reached            */
#define QRGPU_ST_MPC_INFEAS    0x2    /* no feasible step for a violated row, or the solve ended with a row it had set aside as
                                         dependent violated, or an active row not tight, by more than 1e-4 N: the forces are not the optimum */
#define QRGPU_ST_MPC_OVERFLOW  0x4    /* working set outgrew its LDS allotment       */
#define QRGPU_ST_MPC_NOTSPD    0x8    /* Hessian pivot <= 0                          */
#define QRGPU_ST_BAD_TYPE      0x01000000  /* d_type_id names a type outside [0, QRGPU_MAX_TYPES) or one that was never set up: the robot was
                                         computed with the first type that was set up and its result must not be used */
#define QRGPU_ST_PIPE_TIMEOUT  0x02000000  /* pipelined tick: the WBC never saw this robot's MPC forces arrive; the torques must not be used */
#define QRGPU_ST_WBC_MAXITER   0x10
#define QRGPU_ST_WBC_INFEAS    0x20
#define QRGPU_ST_VMC_MAXITER   0x40
#define QRGPU_ST_VMC_INFEAS    0x80   /* QuadProg++ would have returned +inf (e.g. the 1e-7 window of a swing foot, qr_qp_torque_optimizer.cpp:79-81);
                                         the force is the iterate QuadProg++ stops at, which is what the reference goes on to use (:281-297) */

/* What BuildDynamicModel reads from YAML plus what the WBC controller hard-codes.
 * Defaults (qrgpu_model_desc_default) are the A1 values. */
typedef struct {
    float hip_l, upper_l, lower_l;        /* robot_params.hip_l / upper_l / lower_l          */
    float body_size[3];                   /* robot_params.body_size (unused by the torque path) */
    float kp_body_pos, kd_body_pos;       /* 100 / 10   qr_wbc_locomotion_controller.cpp:62-63 */
    float kp_body_ori, kd_body_ori;       /* 100 / 10   :66-67 */
    float kp_foot, kd_foot;               /* 500 / 10   :70-71 */
    float weight_fb, weight_fr;           /* 0.1 / 1    :44-45 */
    float mu;                             /* 0.4        QS/controllers/wbc/qr_single_contact.cpp:34 */
} qrgpu_model_desc;

void qrgpu_model_desc_default(qrgpu_model_desc *d);

/* Per-type constants of the force-balance (VMC) QP: ComputeContactForce's non-per-tick arguments.  Defaults: A1 + the header defaults
 * (QI/controllers/balance_controller/qr_qp_torque_optimizer.h:144-153), acc_weight of config/a1_sim/stance_leg_controller.yaml. */
typedef struct {
    float mass;                           /* robot->totalMass */
    float inertia[9];                     /* robot->totalInertia, Eigen column-major (the YAML list as mapped by MatrixXf::Map) */
    float acc_weight[6];
    float reg_weight, friction, fmin_ratio, fmax_ratio;   /* 1e-4, 0.5, 0.01, 10 */
    float hip_l, upper_l, lower_l;        /* leg geometry for the J^T f torque map */
} qrgpu_vmc_desc;
void qrgpu_vmc_desc_default(qrgpu_vmc_desc *d);

/* ---- lifetime -------------------------------------------------------------- */
/* device_id: HIP device ordinal.  max_batch: largest n of any batched call. */
int  qrgpu_create(int device_id, int max_batch, int horizon_max, qrgpu_ctx **out);
void qrgpu_destroy(qrgpu_ctx *ctx);
/* The compute stream: every batched call is queued on it and completes in its order.  A context starts with a non-blocking stream of its own
 * (qrgpu_get_stream hands it out, for a caller that wants to queue its own work in order with the library's); qrgpu_set_stream names another
 * hipStream_t -- NULL = the default (null) stream, on which several contexts of one process serialise each other's launches. */
int  qrgpu_set_stream(qrgpu_ctx *ctx, void *hip_stream);
void *qrgpu_get_stream(qrgpu_ctx *ctx);
/* Longest-first dispatch of the batched MPC solve (default on): robot i's solve time in one call orders the workgroup
 * dispatch of the next call with the same n (control ticks are temporally coherent).  Scheduling only -- results do not
 * depend on it; a batch whose robots were reshuffled between calls merely loses the speed-up.  Turning it on or off
 * forgets the history. */
int  qrgpu_set_lpt_schedule(qrgpu_ctx *ctx, int on);
/* Warm start of the batched MPC solve (default on): robot slot i's final working set in one call (by leg-step, realigned to the scrolling
 * contact table) is where its solve starts in the next call with the same n -- the equality-constrained problem on that set is solved in
 * one block instead of adding its rows one active-set iteration at a time; rows that no longer belong leave, missing ones are added.  The
 * reference cold-starts qpOASES at every solve (QProblem::init, qr_mpc_interface.cpp:430); the QP has one optimum, so results agree
 * with a cold start to the solver's tolerance (1e-9 relative) -- not bit for bit: switch it off where run-to-run bit identity across a
 * changing history matters.  Turning it on or off forgets what is stored. */
int  qrgpu_set_warm_start(qrgpu_ctx *ctx, int on);
/* Planned list of the batched MPC solve (default on, big_nls = 0): a robot that needed the rescue pass in one call -- or ended within a few
 * rows of the main pass's LDS allotment -- is solved in the NEXT call with the same n by a list launch of its own (whole CU's LDS, 96
 * working-set positions), issued at the start of the call on a second stream of the context beside the main launch, which skips it: the
 * batch no longer waits for a re-solve after the main launch.  big_nls > 0 additionally sends every robot with at least that many stance
 * leg-steps (4h = all feet down over the whole horizon) there.  Scheduling -- with two things a caller may see: the list launches are other
 * instantiations of the kernel than the main pass (whole CU's LDS, 96 working-set positions), so a robot's forces agree between the two to the
 * solver's tolerance (1e-6 of the largest force; the QP has one optimum), not bit for bit; and a robot at the very edge of what a launch can
 * hold may carry QRGPU_ST_MPC_OVERFLOW under one schedule and not under the other (tests/test_gpu_wbc.py allows two such robots in 1024 between
 * the serial and the pipelined tick).  A robot the main pass solves under both schedules has the same bits.
 * The host learns the list's length through pinned memory without a sync; so that a caller which queues calls faster than the GPU runs
 * them still gets its first plans, the first two batched calls after the history was reset (a new n, qrgpu_set_lpt_schedule) end with a
 * hipStreamSynchronize on the context's stream.  Every later call stays asynchronous.
 * At h > 11 and 3.5 robots per CU or more the main pass runs two workgroups per CU on half the LDS each, and the list is also where the robots
 * go whose inverse Hessian does not fit half a CU (43 stance leg-steps and more at h = 16; a smaller big_nls of the caller's stands) and those
 * whose solves are the longest of the tick; with the planned list, the rescue pass or the longest-first schedule switched off such batches
 * run one workgroup per CU.  The same two caveats apply. */
int  qrgpu_set_planned_list(qrgpu_ctx *ctx, int on, int big_nls);
/* Rescue pass of the batched MPC solve (default on).  A working set that outgrows the 64 lanes of the four-wave loop is handed over
 * in place to the single-wave loop (up to 96 rows) -- that needs no switch.  What remains are robots limited by LDS (an all-stance inverse
 * Hessian at h = 10 leaves room for 56 rows): they are re-solved by the same kernel with the whole CU's LDS in a second, normally
 * empty launch, at most 64 robots per call (the rest keep QRGPU_ST_MPC_OVERFLOW), h <= 11 and the two-workgroups-per-CU main pass of h > 11 (otherwise at h = 16 such robots keep S^-1 in a global scratch instead).
 * A robot nothing can hold keeps QRGPU_ST_MPC_OVERFLOW. */
int  qrgpu_set_rescue_pass(qrgpu_ctx *ctx, int on);
/* Pipelined tick (default on; batches of 64 robots and more): qrgpu_tick_batch queues its WBC launch on a stream of the context's own
 * BESIDE the MPC launches instead of behind them.  Of a robot's WBC only the relaxation QP at its end reads the MPC's forces, so a WBC
 * workgroup runs dynamics, task set and kinematic projection while its robot's solve is still going and takes the forces when that
 * solve raises the robot's flag; the call's outputs are complete when the context stream's work is (stream order, as before).  Scheduling
 * only: results are those of the serial form bit for bit.  A WBC workgroup that waits longer than 4 ms for its robot (never observed)
 * gives the robot QRGPU_ST_PIPE_TIMEOUT. */
int  qrgpu_set_tick_pipeline(qrgpu_ctx *ctx, int on);
/* Overlapped ticks (default OFF; pipelined ticks): a caller that queues qrgpu_tick_batch calls without waiting for them may let
 * tick t + 1's solves start in the slots tick t's drain leaves empty -- a quarter of a 1024-robot tick's slot-time -- instead of behind tick
 * t's last workgroup.  With the mode on, a tick's launches go on stream sets of the context's own (two, alternating) and the context's stream
 * carries only the tick's join: OUTPUTS ARE COMPLETE IN CALL ORDER ON THE CONTEXT'S STREAM exactly as before (whatever is queued there behind
 * the call sees them), and results are those of the serial tick bit for bit (a robot that goes through a list launch under one schedule and not
 * under the other: to the solver's tolerance, qrgpu_set_planned_list) -- what a robot carries from tick to tick (warm-start words,
 * smoothed cost, the orientation task's memory prev_ori) is handed from its tick-t workgroup to its tick-(t + 1) workgroup behind per-robot
 * epoch words with bounded waits (20 ms; a robot whose wait gives up starts cold and carries QRGPU_ST_PIPE_TIMEOUT).
 * What the caller promises in exchange -- the mode is another contract than "stream order" for the INPUTS:
 *   (1) a tick's input arrays are complete when the call is made, or were produced by work queued on the context's stream BEFORE THE PREVIOUS
 *       qrgpu_tick_batch call: a chained tick does not wait for work queued on the context's stream since then.  Inputs produced there later
 *       (an upload, a front-end kernel) need qrgpu_tick_fence() in front of the tick, which makes that one tick wait for the whole stream;
 *   (2) consecutive ticks write DIFFERENT output arrays (force, tau, qdes, status: double-buffer them) and share prev_ori.  The library
 *       checks this: a tick that reuses any output array of its predecessor, or changes n or prev_ori, or follows any other launch of this
 *       context, is simply not chained -- it waits for the context's stream (an event) and runs as the pipelined tick always did.
 * Throughput, not latency: a chained tick completes LATER after its call than an unchained one (its join waits for a launch that shares the
 * machine with the next tick); bench.py prints both beside `value` (config.tick_latency_ms, config.ticks_per_s_no_tick_overlap).
 * Call it AFTER the context's types are set up: the stream sets are made for the horizon the context has at the call (h <= 11: two lanes on the
 * whole machine; h > 11, from 3.5 robots per CU on: two lanes on a machine split by CU masks -- the main pass two to a CU on 192 CUs, the robots
 * that need a whole CU on 64 reserved ones); a context whose horizon changes class afterwards runs plain pipelined ticks until
 * the call is made again.  Throughput of the h > 11 form shows on sequences, not on a handful of ticks: a tick completes about 1.7 periods after
 * its call (1.74 against 1.51 M ticks/s on the mixed h = 16 shard over 25-tick windows, par over 5-tick windows).
 * Needs the context's streams on hardware queues of their own: qrgpu_set_tick_overlap(1) probes that and returns QRGPU_ERR_NOT_SETUP (mode
 * stays off, qrgpu_last_error says why) when two of them share one -- set GPU_MAX_HW_QUEUES=8 in the environment before the process's
 * first HIP call (the HIP runtime's default of 4 is fewer than the streams a context owns). */
int  qrgpu_set_tick_overlap(qrgpu_ctx *ctx, int on);
int  qrgpu_tick_fence(qrgpu_ctx *ctx);
/* How many ticks of this context ran on the overlapped form so far: chained to their predecessor / behind an event of the context's stream. */
int  qrgpu_tick_overlap_stats(const qrgpu_ctx *ctx, int *chained, int *unchained);
const char *qrgpu_last_error(const qrgpu_ctx *ctx);
/* Device facts for reports: returns CU count, writes name (<= len). */
int  qrgpu_device_info(const qrgpu_ctx *ctx, char *name, int len, int *lds_per_cu_bytes);

/* ---- setup (SetupProblem / BuildDynamicModel) -------------------------------
 * A context has ONE horizon (the reference has one global problem size): setting a type up with a different
 * horizon invalidates the MPC setup of the other types until they are set up again with the new horizon. */
int qrgpu_mpc_setup(qrgpu_ctx *ctx, int type_id, float dt, int horizon, float mu, float fmax, float mass,
                    const float inertia[3], const float weights[12], float alpha);
int qrgpu_wbc_setup(qrgpu_ctx *ctx, int type_id, const qrgpu_model_desc *desc);

/* Arithmetic of the Hessian contraction qH = temp * Bqp (K4, qr_mpc_interface.cpp:396-412) of every MPC solve of this context:
 *   QRGPU_HESSIAN_F32     (default) v_mfma_f32_16x16x4_f32: bit for bit the k-ordered fp32 fmaf chain of the reference's dense GEMM;
 *   QRGPU_HESSIAN_BF16X3  BASELINE.json configs[4] ("fp32 QP + bf16 Hessian MFMA"): every fp32 operand cut into three bf16 limbs, the six
 *                         leading cross products per term summed in fp32 by v_mfma_f32_16x16x32_bf16.  H comes out within an ulp or two of the
 *                         exact fp32 assembly (measured 7.5e-9 absolute: the size of the exact H's own asymmetry) but not bit-identical, and on
 *                         this QP an ulp of H is amplified by 1 / (2 alpha), so the mode states ANOTHER QP, a rounding away, and solves that
 *                         one exactly: forces within 1e-5 max(1, |f|) and full-tick torque within 1e-4 max(1, |tau|) of the CPU oracle's solve
 *                         of the very (H, g) this mode assembles, every robot of the configs[4] per-GPU shard
 *                         (tests/test_gpu_mpc.py::test_bf16x3_solve_vs_oracle_on_its_own_hessian).  Against the default mode's answer the
 *                         forces move as the reference's own do between H and H^T.  The default stays QRGPU_HESSIAN_F32 for every
 *                         configuration, configs[4] included: it is bit-exact against the reference's fp32 GEMM AND faster here (the limb
 *                         cuts dominate the operand generation); this mode exists because that configuration names it. */
#define QRGPU_HESSIAN_F32    0
#define QRGPU_HESSIAN_BF16X3 1
int qrgpu_mpc_set_hessian_mode(qrgpu_ctx *ctx, int mode);

/* ---- batched device-pointer API (the measured path) -------------------------- */
/* d_type_id may be NULL (all robots type 0).  d_q: joint angles [12][n] (needed by the
 * J^T f torque map; pass fb_state + 13*n to reuse the WBC state).  d_status may be NULL. */
int qrgpu_mpc_solve_batch(qrgpu_ctx *ctx, int n, const int *d_type_id, const float *d_mpc_state,
                          const float *d_traj, const float *d_gait, const float *d_q,
                          float *d_force, float *d_tau_mpc, int *d_status);
int qrgpu_wbc_run_batch(qrgpu_ctx *ctx, int n, const int *d_type_id, const float *d_fb_state,
                        const float *d_wbc_cmd, float *d_prev_ori, float *d_tau,
                        float *d_qdes /* [24][n]: desiredJPos, desiredJVel; may be NULL */, int *d_status);
/* Full tick: MPC, then WBC with Fr_des := that MPC's forces (the Fr_des rows of d_wbc_cmd are ignored).
 * d_tau receives the WBC torque on stance legs and the MPC J^T f torque on swing legs
 * (UpdateLegCMD only overwrites stance legs, qr_wbc_locomotion_controller.cpp:205-219).
 * d_qdes [24][n] (may be NULL): desiredJPos, desiredJVel of the kinematic multitask projection (K12, qrMultitaskProjection::FindConfiguration,
 * which qrWbcLocomotionController::Run always executes, qr_wbc_locomotion_controller.cpp:124-129); with NULL that projection is skipped
 * (its outputs do not enter the torque). */
int qrgpu_tick_batch(qrgpu_ctx *ctx, int n, const int *d_type_id, const float *d_mpc_state,
                     const float *d_traj, const float *d_gait, const float *d_fb_state,
                     const float *d_wbc_cmd, float *d_prev_ori, float *d_force, float *d_tau, float *d_qdes, int *d_status);

/* Motor-command tail of the locomotion state (K14), off by default -- the single-robot drop-in keeps the FSM doing it:
 *   QRGPU_EPILOGUE_HIP_COMP  the +-0.9 N m abad compensation of qrFSMStateLocomotion::Run (QS/fsm/qr_fsm_state_locomotion.cpp:141-151:
 *                            legs FR, RR -0.9, legs FL, RL +0.9, added to every leg's command BEFORE the WBC overwrites its stance legs,
 *                            so in the full tick it survives on swing legs only; in qrgpu_mpc_solve_batch on every leg)
 *   QRGPU_EPILOGUE_CLIP      the +-23 N m clip of qrSafetyChecker::CheckForceFeedForward (QS/fsm/qr_safety_checker.cpp:48-66)
 * Applies to d_tau of qrgpu_tick_batch and qrgpu_mpc_solve_batch. */
#define QRGPU_EPILOGUE_HIP_COMP 1
#define QRGPU_EPILOGUE_CLIP     2
int qrgpu_set_torque_epilogue(qrgpu_ctx *ctx, int flags);

/* MPC front-end of n robots for one control tick: MPCStanceLegController::SetupCommand + Run + UpdateMPC without the solve
 * (QS/controllers/mpc/qr_mpc_stance_leg_controller.cpp:158-204, 207-334, 337-382).  Writes the contact table d_gait every tick,
 * the reference trajectory d_traj only for robots that re-plan this tick (iterationCounter % (round(dt_mpc/dt_ctrl)/2) == 0
 * or < 50; d_mpc_updated[i] = 1, may be NULL), and rows 0-14 and 63-66 of d_wbc_cmd (may be NULL): pBody_des, vBody_des,
 * aBody_des = 0, pBody_RPY_des, vBody_Ori_des, contact_state.  The horizon is the one of qrgpu_mpc_setup.
 * Reference values: dt_ctrl 0.002, dt_mpc 0.06, num_horizon_l = max(2, int(fullCyclePeriod/0.4)) (:43-50). */
int qrgpu_mpc_frontend_batch(qrgpu_ctx *ctx, int n, int num_horizon_l, float dt_ctrl, float dt_mpc, const float *d_fe_in,
                             float *d_fe_state, float *d_traj, float *d_gait, float *d_wbc_cmd, int *d_mpc_updated);

/* Force-balance stance forces of n robots: contact forces in the base frame, force[3*leg+axis] (the 3x4 matrix ComputeContactForce
 * returns, column-major) and, when d_q and d_tau are given, the joint torques J^T f of MapContactForceToJointTorques. */
int qrgpu_vmc_setup(qrgpu_ctx *ctx, int type_id, const qrgpu_vmc_desc *desc);
int qrgpu_vmc_force_batch(qrgpu_ctx *ctx, int n, const int *d_type_id, const float *d_vmc_in, const float *d_q /*[12][n], may be NULL*/,
                          float *d_force, float *d_tau /*may be NULL*/, int *d_status /*may be NULL*/);
/* The world-frame overload of ComputeContactForce (qr_qp_torque_optimizer.cpp:304-398; what TorqueStanceLegController::GetAction calls when
 * user_parameters.yaml computeForceInWorldFrame is true, qr_torque_stance_leg_controller.cpp:490-498): the same QP with, in d_vmc_in,
 * Rcb := rotMat (base -> world, quaternionToRotationMatrix(quat)^T), gvec := (0, 0, 9.8), normal := (0, 0, 1) -- the tangents are then the
 * world x and y axes, as GetAction passes them -- and per-leg force-window ratios d_ratio [8][n] = fMinRatio[4], fMaxRatio[4] (the Vec4
 * members that the walk mode changes per tick, :128-152).  Forces come back in the base frame, as RigidTransform(0, quat, X^T) returns them. */
int qrgpu_vmc_force_world_batch(qrgpu_ctx *ctx, int n, const int *d_type_id, const float *d_vmc_in, const float *d_ratio, const float *d_q,
                                float *d_force, float *d_tau, int *d_status);

/* Base velocity estimator (TinyEKF<3,3> + moving-window filters), the leg kinematics it reads and the pose estimator that follows it
 * (stance-foot height, planar odometry), for n robots and one control tick.
 * d_est_state is the estimators' memory: qrgpu_estimator_state_doubles(window) doubles per robot, [field][robot], zero = freshly
 * constructed / Reset() (qr_robot_velocity_estimator.cpp:29-60).  d_tick: robot->GetTick() in milliseconds.  All robots share `desc`. */
typedef struct {
    float hip_l, upper_l, lower_l;
    float hip_offset[12];                 /* GetDefaultHipPosition, 3*leg+axis */
    float time_step;                      /* robot->timeStep, used for the first sample */
    float accelerometer_variance, sensor_variance;   /* user_parameters.yaml: 0.1, 0.1 */
    int window;                           /* movingWindowFilterSize: 120 */
    float body_height;                    /* robot->bodyHeight, the height reported while no foot is in stance */
} qrgpu_estimator_desc;
void qrgpu_estimator_desc_default(qrgpu_estimator_desc *d);
int qrgpu_estimator_state_doubles(int window);
#define QRGPU_EST_IN_ROWS  54     /* rows of est_in  */
#define QRGPU_EST_OUT_ROWS 42     /* rows of est_out */
int qrgpu_estimator_update_batch(qrgpu_ctx *ctx, int n, const qrgpu_estimator_desc *desc, const float *d_est_in, const unsigned *d_tick,
                                 double *d_est_state, float *d_est_out);

/* Ground-plane fit and control frame of n robots for one control tick: qrGroundSurfaceEstimator::Update / GetNormalVector /
 * ComputeControlFrame (QS/estimators/qr_ground_surface_estimator.cpp:40-70,151-206), which qrStateEstimatorContainer::Update runs in front
 * of the robot estimator (QI/estimators/qr_state_estimator_container.h:76-81).  The fit fires when all four feet are in contact and one of
 * them newly so; the control frame is the base's heading (the reference assumes a flat ground in the world, :168), low-pass filtered as
 * roll / pitch / yaw with ratio 0.8 and roll := 0.
 * d_ground_in [23][n]: footContact[4], footPositionsInBaseFrame[12] (3*leg+axis), basePosition[3], quat_wxyz[4].
 * d_ground_state [QRGPU_GROUND_STATE_DOUBLES][n] doubles is the estimators' memory (lastContactState[4], a[3], n[3], controlFrameRPY[3]);
 * reset != 0 applies Reset() before the update.  d_ground_out [QRGPU_GROUND_OUT_ROWS][n] (may be NULL): plane coefficients a[3]
 * (z = a0 + a1 x + a2 y, base frame), unit normal n[3], controlFrameRPY[3], controlFrameOrientation[4], stateDataFlow.groundRMat[9]
 * (row-major), stateDataFlow.baseRInControlFrame[9], updated flag.  d_est_in (may be NULL): the estimator's input array, whose rows 45-53
 * (GetAlignedDirections) receive groundRMat, so that ground fit -> pose estimator stays on the device. */
#define QRGPU_GROUND_IN_ROWS 23
#define QRGPU_GROUND_STATE_DOUBLES 13
#define QRGPU_GROUND_OUT_ROWS 32
int qrgpu_ground_update_batch(qrgpu_ctx *ctx, int n, int reset, const float *d_ground_in, double *d_ground_state, float *d_ground_out,
                              float *d_est_in);

/* Open-loop gait generator (qrOpenLoopGaitGenerator::Update + Schedule, QS/gait/qr_openloop_gait_generator.cpp:126-249) of n robots for
 * one control tick.  d_contact [4][n]: robot->GetFootContact().  d_gait_state [QRGPU_GAIT_STATE_FLOATS][n] is the generators' memory;
 * reset != 0 applies Reset(0) before the update.  d_gait_out [24][n] (may be NULL): phaseInFullCycle[4], normalizedPhase[4],
 * desiredLegState[4], legState[4], curLegState[4], swingTimeRemaining[4].  d_fe_in (may be NULL): the front-end's input array, whose rows
 * 42-61 (phaseInFullCycle, dutyFactor, normalizedPhase, desiredLegState, legState) are written.  Legs with duty factor 0
 * (USERDEFINED_SWING) are not supported. */
#define QRGPU_GAIT_STATE_FLOATS 52
typedef struct {
    float stance_duration[4], duty_factor[4], initial_leg_phase[4];   /* openloop_gait_generator.yaml: 0.5, 0.6, (0.5, 0, 0, 0.5) for advanced_trot */
    int initial_leg_state[4];                                          /* LegState: SWING 0, STANCE 1 */
    float contact_detection_phase_threshold, wait_time;                /* 0.5, gait_params.wait_time */
    int advanced_trot;                                                 /* gait == "advanced_trot": the lost-contact hold of Schedule() */
} qrgpu_gait_desc;
void qrgpu_gait_desc_default(qrgpu_gait_desc *d);
int qrgpu_gait_update_batch(qrgpu_ctx *ctx, int n, const qrgpu_gait_desc *desc, float current_time, int robot_stop, int reset, const float *d_contact,
                            float *d_gait_state, float *d_gait_out, float *d_fe_in);

/* Walk gait generator of n robots for one control tick (qrWalkGaitGenerator::Update, QS/gait/qr_walk_gait_generator.cpp:202-288; the
 * constructor's bookkeeping :66-193 is done on the host from `desc`), and the contacts / force-window ratios its sub-states select in
 * TorqueStanceLegController::UpdateFRatio's walk branch (QS/controllers/balance_controller/qr_torque_stance_leg_controller.cpp:125-168).
 * current_time is the time the reference passes to Update (it is NOT taken relative to the last reset there).  d_contact [4][n]:
 * robot->GetFootContact().  d_walk_state [QRGPU_WALK_STATE_FLOATS][n] is the generators' memory; reset: 2 = as constructed, 1 = Reset()
 * (stateIndexOfLegs and the detection members survive a Reset in the reference), 0 = carry on.
 * d_walk_out [QRGPU_WALK_OUT_ROWS][n] (may be NULL): phaseInFullCycle[4], normalizedPhase[4], desiredLegState[4] (LegState /
 * SubLegState values: STANCE 1, LOAD_FORCE 5, UNLOAD_FORCE 6, FULL_STANCE 7, TRUE_SWING 8), legState[4], curLegState[4],
 * detectedLegState[4] (SWING 0, STANCE 1, EARLY_CONTACT 2, LOSE_CONTACT 3), detectedEventTickPhase[4], moveBasePhase, contacts[4],
 * fMinRatio[4], fMaxRatio[4].  d_ratio [8][n] (may be NULL): fMinRatio, fMaxRatio laid out as qrgpu_vmc_force_world_batch takes them;
 * d_vmc_in (may be NULL): its rows 18-21 (contacts) are written.  Legs with duty factor 0 (USERDEFINED_SWING) are not supported; members the
 * reference leaves uninitialised until their first write (moveBasePhase, detectedLegState, detectedEventTickPhase) start at zero. */
#define QRGPU_WALK_STATE_FLOATS 33
#define QRGPU_WALK_OUT_ROWS 41
typedef struct {
    float stance_duration[4], duty_factor[4], initial_leg_phase[4];   /* openloop_gait_generator.yaml, gait "walk": 7.5, 0.75, (0.5, 0, 0.75, 0.25) */
    int initial_leg_state[4];                                          /* LegState: SWING 0, STANCE 1 */
    float contact_detection_phase_threshold;                           /* 0.1 */
    int n_states;                                                      /* entries of state_switch_que / state_ratio (<= 4) */
    int state_switch[4];                                               /* SubLegState: full_stance 7, unload_force 6, true_swing 8, load_force 5 */
    float state_ratio[4];                                              /* 0.2, 0.3, 0.3, 0.2 */
} qrgpu_walk_gait_desc;
void qrgpu_walk_gait_desc_default(qrgpu_walk_gait_desc *d);
int qrgpu_walk_gait_update_batch(qrgpu_ctx *ctx, int n, const qrgpu_walk_gait_desc *desc, float current_time, int robot_stop, int reset,
                                 const float *d_contact, float *d_walk_state, float *d_walk_out, float *d_ratio, float *d_vmc_in);

/* Swing-leg targets of the MPC/WBC mode (qrRaibertSwingLegController::GetAction, ADVANCED_TROT case on horizontal terrain,
 * QS/controllers/qr_swing_leg_controller.cpp:362-398,408-424): XY-linear / Z-parabola foot trajectory between the lift-off point and the
 * planned foothold, its world-frame image for the WBC foot tasks, and the inverse-kinematics joint targets of the swing command.
 * swing_in [58][n]: swing flag[4], footholdPlanner phase[4], swingDuration[4], phaseSwitchFootGlobalPos[12], desiredFootholds[12] (base
 * frame), basePosition[3], quat_wxyz[4], baseVInWorldFrame[3], motor angles[12].  For the flagged legs only: rows 15-50 of d_wbc_cmd
 * (pFoot_des, vFoot_des, aFoot_des), d_foot_target_world [12][n] (footTargetPositionsInWorldFrame, rows 26-37 of fe_in) and
 * d_qdes [24][n] (joint angle and velocity targets).  Any output may be NULL.  desc: leg lengths and hip offsets. */
int qrgpu_swing_targets_batch(qrgpu_ctx *ctx, int n, const qrgpu_estimator_desc *desc, const float *d_swing_in, float *d_wbc_cmd,
                              float *d_foot_target_world, float *d_qdes);

/* Swing-leg action of the velocity mode -- the trot that the force-balance path (qrgpu_vmc_force_batch) belongs to:
 * qrRaibertSwingLegController::GetAction, VELOCITY_LOCOMOTION case (QS/controllers/qr_swing_leg_controller.cpp:285-309, 408-424): the
 * Raibert target in the base frame from the hip's horizontal velocity, XYLinear_ZParabola (height 0.1) between the lift-off point and that
 * target at the warped phase of GenerateTrajectoryPoint(phaseModule = true), leg inverse kinematics.
 * swing_vel_in [53][n]: swing flag[4] (the leg is in swingFootIds), normalizedPhase[4], phaseSwitchFootLocalPos[12] (3*leg+axis), estimated
 * base velocity in the base frame[3], yaw rate, desiredSpeed[3] (stateDes 6..8), desiredTwistingSpeed (stateDes 11), dR[9]
 * (stateDataFlow.baseRInControlFrame, row-major: rows 22-30 of qrgpu_ground_update_batch's output; the identity on PLANE / PLUM_PILES
 * terrain), quat_wxyz[4], motor angles[12].  d_out [48][n], for the flagged legs only: footTargetPosition[12] (base frame),
 * footPositionInBaseFrame[12], joint angle targets[12], joint velocity targets[12] (zero: the reference's parabola generator returns no
 * velocity).  desc: leg lengths and robot->hipOffset (the inverse kinematics); vdesc: the controller's own parameters. */
typedef struct {
    float hip_position_com[12];       /* GetDefaultHipPosition() + comOffset, 3*leg+axis                              */
    float stance_duration[4];         /* gaitGenerator->stanceDuration                                                */
    float swing_kp[3];                /* user_parameters.yaml swingKp.trot: 0.03 x3                                   */
    float desired_height;             /* user_parameters desiredHeight - footClearance: 0.26                          */
} qrgpu_swing_velocity_desc;
int qrgpu_swing_velocity_batch(qrgpu_ctx *ctx, int n, const qrgpu_estimator_desc *desc, const qrgpu_swing_velocity_desc *vdesc,
                               const float *d_swing_vel_in, float *d_out);

/* Swing-leg selection and the Raibert-type foothold heuristic that feed qrgpu_swing_targets_batch: qrRaibertSwingLegController::Update
 * (default branch, QS/controllers/qr_swing_leg_controller.cpp:211-236) + qrFootholdPlanner::ComputeHeuristicFootHold
 * (QS/planner/qr_foothold_planner.cpp:110-239) on flat ground (groundRMat = I).
 * fh_in [46][n]: legState[4] (LegState: SWING 0, STANCE 1, EARLY_CONTACT 2, LOSE_CONTACT 3), allowSwitchLegState[4], swingTimeRemaining[4],
 * normalizedPhase[4], desiredSpeed[3] (stateDes 6..8), desiredTwistingSpeed (stateDes 11), stateDes(2), footPositionsInBaseFrame[12],
 * quat_wxyz[4], rpy[3], baseVelocityInBaseFrame[3], baseRollPitchYawRate[3].  When d_gait_state / d_gait_out (the arrays of
 * qrgpu_gait_update_batch) are given, rows 0-15 are taken from them instead.  Writes swing_in rows 0-3 (leg is in swingFootIds) for every
 * leg and, for those legs, rows 4-7 (footholdPlanner->phase) and 24-35 (desiredFootholds, base frame); other rows are left alone.
 * The reference's out-of-bounds write for backwards commands (:222-225, footTargetPosition(0,2) on a 3x1 vector) is not reproduced. */
typedef struct {
    float hip_offset[12];             /* robot->hipOffset, 3*leg+axis (robot_params.hip_offset)                      */
    float default_hip_position[12];   /* robot->GetDefaultHipPosition() (robot_params.default_hip_positions)         */
    float hip_l;                      /* robot->hipLength                                                             */
    float swing_kp[3];                /* user_parameters.yaml swingKp.advanced_trot: 0.16 x3                          */
    float foot_clearance;             /* user_parameters.yaml footClearance: 0.01                                     */
} qrgpu_foothold_desc;
void qrgpu_foothold_desc_default(qrgpu_foothold_desc *d);
int qrgpu_footholds_batch(qrgpu_ctx *ctx, int n, const qrgpu_foothold_desc *desc, const float *d_fh_in, const float *d_gait_state,
                          const float *d_gait_out, float *d_swing_in);

/* The tick's state arrays from the estimator's inputs and outputs: what SolveDenseMPC (qr_mpc_stance_leg_controller.cpp:385-399:
 * pos, baseVInWorldFrame, quat, baseWInWorldFrame, foot2ComInWorldFrame = baseRMat (footPositionsInBaseFrame - comOffset), rpy) and
 * qrWbcLocomotionController::UpdateModel (qr_wbc_locomotion_controller.cpp:136-156) read.  d_rpy [3][n] = GetBaseRollPitchYaw.
 * Either output may be NULL. */
int qrgpu_pack_state_batch(qrgpu_ctx *ctx, int n, const float com_offset[3], const float *d_est_in, const float *d_est_out, const float *d_rpy,
                           float *d_mpc_state, float *d_fb_state);

/* ---- multi-GPU: all-gather of the per-robot torques over RCCL / xGMI ----------------------------------------
 * One process per GPU, one context per process; rank r of N owns a contiguous shard of the robot population and there is no exchange
 * inside a tick.  The only collective of the path collects every rank's tau[12][n_local] on every rank:
 *     d_tau_all [nranks][12][n_local]   (rank-major; block r is rank r's d_tau)
 * The communicator is created from a 128-byte id blob (ncclUniqueId) that rank 0 makes and the launcher hands to the other ranks by
 * whatever channel it has (MPI, a file, torch.distributed's CPU backend ...): no collective library appears in this ABI's signatures.
 * qrgpu_allgather_tau takes either that context-owned communicator (nccl_comm = NULL) or the caller's own ncclComm_t.  It is asynchronous:
 * the gather waits for the work queued on the context's compute stream so far, runs on a stream of the context's own, and the next
 * tick may be issued at once.  `slot` (0 / 1) names the torque buffer being read for double buffering: call qrgpu_allgather_fence(slot)
 * before queueing work that overwrites that buffer, qrgpu_comm_sync to wait on the host for every gather issued so far.
 * A consumer of d_tau_all queued on the context's compute stream waits for the gather with qrgpu_allgather_wait(slot) (a stream-side wait,
 * no host block; it leaves the fence of that slot due).  The gather of tick i + 1 starts as soon as tick i + 1's torques are complete: a
 * consumer of tick i's d_tau_all that is queued AFTER qrgpu_allgather_tau of tick i + 1 would race with it, so either queue the consumer
 * first or give d_tau_all two buffers by slot as well. */
#define QRGPU_COMM_ID_BYTES 128
int qrgpu_comm_unique_id(unsigned char id[QRGPU_COMM_ID_BYTES]);
int qrgpu_comm_init_rank(qrgpu_ctx *ctx, const unsigned char id[QRGPU_COMM_ID_BYTES], int nranks, int rank);
int qrgpu_comm_info(const qrgpu_ctx *ctx, int *nranks, int *rank);
int qrgpu_comm_destroy(qrgpu_ctx *ctx);
int qrgpu_allgather_tau(qrgpu_ctx *ctx, void *nccl_comm /* ncclComm_t, or NULL = the context's */, const float *d_tau /* [12][n_local] */,
                        int n_local, float *d_tau_all /* [nranks][12][n_local] */, int slot);
/* The same for the torque array that the context's most recent qrgpu_tick_batch produced, when nothing the caller has queued on the context's
 * stream since writes it: the gather then waits for THAT TICK to be complete (its join bumps a count the communication stream polls) instead of for
 * an event of the context's stream -- recording one costs that stream several microseconds a tick.  Falls back to qrgpu_allgather_tau's form when
 * the last tick was not a pipelined one (fewer than 64 robots, qrgpu_set_tick_pipeline off, a stream under graph capture). */
int qrgpu_allgather_tau_of_tick(qrgpu_ctx *ctx, void *nccl_comm, const float *d_tau, int n_local, float *d_tau_all, int slot);
int qrgpu_allgather_fence(qrgpu_ctx *ctx, int slot);
int qrgpu_allgather_wait(qrgpu_ctx *ctx, int slot);
int qrgpu_comm_sync(qrgpu_ctx *ctx);

/* ---- single-robot host-pointer API (what the drop-in C++ adapters call) ------ */
int qrgpu_mpc_solve1(qrgpu_ctx *ctx, int type_id, const float p[3], const float v[3], const float quat_wxyz[4],
                     const float w[3], const float r_3x4_colmajor[12], const float rpy[3],
                     const float *traj /*12h*/, const float *gait /*4h*/, const float q[12] /*may be NULL*/,
                     double f_out[12], float tau_out[12] /*may be NULL*/, int *status);
int qrgpu_wbc_run1(qrgpu_ctx *ctx, int type_id, const float fb_state[37], const float wbc_cmd[67],
                   float prev_ori_vel[3], float tau_out[12], float qdes_out[12], float qddes_out[12], int *status);

int qrgpu_vmc_force1(qrgpu_ctx *ctx, int type_id, const float vmc_in[37], const float q[12] /*may be NULL*/,
                     float force_out[12], float tau_out[12] /*may be NULL*/, int *status);
int qrgpu_vmc_force_world1(qrgpu_ctx *ctx, int type_id, const float vmc_in[37], const float ratio[8] /* fMinRatio[4], fMaxRatio[4] */,
                           const float q[12] /*may be NULL*/, float force_out[12], float tau_out[12] /*may be NULL*/, int *status);

/* ---- inspection (parity tests): the fp32 QP data the MPC kernel assembled ----- */
/* d_H: [n][12h*12h] row-major per robot, d_g: [n][12h]; entries that involve a swing
 * (eliminated) variable are left untouched. */
int qrgpu_mpc_assemble_batch(qrgpu_ctx *ctx, int n, const int *d_type_id, const float *d_mpc_state,
                             const float *d_traj, const float *d_gait, float *d_H, float *d_g);
/* Rigid-body quantities the WBC kernel computed: d_out [n][QRGPU_FB_DEBUG_FLOATS]
 * = H(324) G(18) C(18) Jc(4*54) Jcdqd(12) pGC(12) vGC(12). */
#define QRGPU_FB_DEBUG_FLOATS (324 + 18 + 18 + 216 + 12 + 12 + 12)
int qrgpu_fb_debug_batch(qrgpu_ctx *ctx, int n, const int *d_type_id, const float *d_fb_state, float *d_out);

/* One WBC tick as qrgpu_wbc_run_batch computes it (K12 skipped), plus the solution of its relaxation QP:
 * d_qp [n][QRGPU_WBC_QP_FLOATS] = qpz[18] (z_fb[6], z_f[3 * contacts], zero padded; qr_wholebody_impulse_ctrl.cpp:113) and
 * extraData->optimalFr[12] = z_f + Fr_des (:216-218), stance feet in contact order, zero padded. */
#define QRGPU_WBC_QP_FLOATS 30
int qrgpu_wbc_inspect_batch(qrgpu_ctx *ctx, int n, const int *d_type_id, const float *d_fb_state, const float *d_wbc_cmd,
                            float *d_prev_ori, float *d_tau, float *d_qp, int *d_status);

/* ---- executed arithmetic of the batched MPC solve (measurement; off by default) ---------------------------------------------------
 * With counting on, every solve leaves what it actually computed, by formula from the sizes it saw (stance leg-steps, tile count,
 * working-set size of every change, rebuilds): out[0] fp32 vector flops, out[1] fp32 matrix flops issued (v_mfma_f32_16x16x4_f32),
 * out[2] fp64 flops of the block sweep and x0, out[3] fp64 flops of the active set.  qrgpu_mpc_flop_counts sums them over the robots of
 * the last counted call (mul and add count 1, fma 2).  bench.py's roofline block is built from these, not from a dense yardstick. */
int qrgpu_enable_flop_count(qrgpu_ctx *ctx, int on);
int qrgpu_mpc_flop_counts(qrgpu_ctx *ctx, double out[4]);

/* ---- plumbing ------------------------------------------------------------------ */
int  qrgpu_sync(qrgpu_ctx *ctx);                       /* hipStreamSynchronize on the context stream */
/* Mean device time (ms) of the kernels launched by the last `calls` batched calls,
 * measured with hipEvents on the context stream; kernel: 0 = MPC, 1 = WBC.
 * on = 0: off; 1: events around every launch; N > 1: around every N-th launch of a kernel (an event
 * pair costs a 0.3 ms tick about 4 us per kernel; the mean is over the bracketed launches);
 * -1: pause (the next on > 0 carries on with what was measured so far; on = 0 forgets it). */
int  qrgpu_enable_timing(qrgpu_ctx *ctx, int on);
int  qrgpu_get_timing(qrgpu_ctx *ctx, int kernel, double *mean_ms, int *count);
void *qrgpu_malloc(qrgpu_ctx *ctx, unsigned long long bytes);   /* hipMalloc on the context device */
void qrgpu_free(qrgpu_ctx *ctx, void *p);
int  qrgpu_memcpy_h2d(qrgpu_ctx *ctx, void *dst, const void *src, unsigned long long bytes);
int  qrgpu_memcpy_d2h(qrgpu_ctx *ctx, void *dst, const void *src, unsigned long long bytes);
/* Pinned host memory and asynchronous copies on the context stream (kind: 0 host->device, 1 device->host, 2 device->device; the host
 * side should be qrgpu_host_alloc memory), an asynchronous byte fill, and caller-indexed timing marks: qrgpu_mark(i) records event i on
 * the context stream (index < 65536; created on first use), qrgpu_mark_elapsed_ms waits for mark `to` and returns the device time between
 * the two.  Launcher plumbing for callers that use no GPU array library (bench.py). */
void *qrgpu_host_alloc(qrgpu_ctx *ctx, unsigned long long bytes);
void qrgpu_host_free(qrgpu_ctx *ctx, void *p);
int  qrgpu_memcpy_async(qrgpu_ctx *ctx, void *dst, const void *src, unsigned long long bytes, int kind);
int  qrgpu_memset_async(qrgpu_ctx *ctx, void *dst, int byte_value, unsigned long long bytes);
int  qrgpu_mark(qrgpu_ctx *ctx, int index);
int  qrgpu_mark_elapsed_ms(qrgpu_ctx *ctx, int from, int to, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* QRGPU_H */
