// TEST INFRASTRUCTURE (see qr_oracle.h).
//
// Open-loop gait generator (producer of the front-end's and the swing controller's phase inputs): restates
//   qrGaitGenerator::Reset                      QI/gait/qr_gait.h:76-87
//   qrOpenLoopGaitGenerator::Reset / Update / Schedule   QS/gait/qr_openloop_gait_generator.cpp:77-123, 126-207, 210-249
// for legs with a non-zero duty factor (the USERDEFINED_SWING branch of :91-103 is not restated).  Plain float arithmetic, no
// library calls except fmod (exact), so the HIP kernel is expected to match bit for bit.
#include "qr_oracle.h"

namespace qro {

void gait_reset(const GaitConfig &c, GaitState &s)
{
    s = GaitState();
    for (int l = 0; l < 4; ++l) { s.cur[l] = s.last[l] = s.leg[l] = s.desired[l] = c.initial_leg_state[l]; }
}

// contact[4]: robot->GetFootContact(); stop: robot->stop.  out[24]: phaseInFullCycle, normalizedPhase, desiredLegState, legState,
// curLegState, swingTimeRemaining (4 each).
void gait_update(const GaitConfig &c, float currentTime, const float contact[4], bool stop, GaitState &s, float out[24])
{
    float full[4], swingDur[4], ratio[4];
    for (int l = 0; l < 4; ++l) { full[l] = c.stance_duration[l] / c.duty_factor[l]; swingDur[l] = full[l] - c.stance_duration[l]; ratio[l] = c.duty_factor[l]; }
    float timeSinceReset = currentTime;
    // ---- Schedule (:210-249)
    bool early_return = false;
    if (s.reset_time + full[0] < timeSinceReset) { s.reset_time = timeSinceReset; s.gait_cycle += 1; }
    timeSinceReset -= s.reset_time;
    for (int l = 0; l < 4; ++l) s.allow[l] = 1;
    if (c.advanced_trot) {
        for (int l = 0; l < 4; ++l)
            if (s.cur[l] == 0 /*SWING*/ && s.desired[l] == 1 /*STANCE*/ && contact[l] == 0.f) s.allow[l] = 0;
        if (s.allow[0] + s.allow[1] + s.allow[2] + s.allow[3] < 4) {
            const float dt_ = currentTime - s.last_time;
            s.cum_dt += dt_;
            if (s.cum_dt > c.wait_time) { for (int l = 0; l < 4; ++l) s.allow[l] = 1; early_return = true; }
            if (!early_return) s.reset_time += dt_;
        } else s.cum_dt = 0;
    }
    // ---- Update (:126-207)
    const bool all_allowed = s.allow[0] + s.allow[1] + s.allow[2] + s.allow[3] == 4;
    for (int l = 0; l < 4; ++l) {
        if (!all_allowed) continue;
        if (!stop || (stop && s.last[l] == 0)) { s.last[l] = s.cur[l]; s.cur[l] = s.desired[l]; }
        const float augmented = c.initial_leg_phase[l] * full[l] + timeSinceReset;
        s.phase[l] = std::fmod(augmented, full[l]) / full[l];
        if (s.phase[l] < ratio[l]) { s.desired[l] = 1; s.nphase[l] = s.phase[l] / ratio[l]; }
        else {
            s.desired[l] = 0;
            s.nphase[l] = (s.phase[l] - ratio[l]) / (1 - ratio[l]);
            if (s.cur[l] == 1) { s.first_swing[l] = 1; s.contact_start_phase[l] = 0; s.first_stance[l] = 0; s.swing_remaining[l] = swingDur[l]; }
            else { s.first_swing[l] = 0; s.swing_remaining[l] = swingDur[l] * (1 - s.nphase[l]); }
        }
        if (s.leg[l] == 2 /*EARLY_CONTACT*/ && s.desired[l] == 0) continue;
        s.leg[l] = s.desired[l];
        if (s.nphase[l] < c.contact_detection_phase_threshold) continue;
        if (s.leg[l] == 0 && contact[l] != 0.f) { s.leg[l] = 2; s.contact_start_phase[l] = s.phase[l] - 1.0f; }
        if (s.cur[l] == 0 && (s.leg[l] == 2 || s.leg[l] == 1)) { s.first_stance[l] = 1; s.first_swing[l] = 0; }
    }
    s.last_time = currentTime;
    for (int l = 0; l < 4; ++l) {
        out[l] = s.phase[l]; out[4 + l] = s.nphase[l]; out[8 + l] = (float)s.desired[l]; out[12 + l] = (float)s.leg[l];
        out[16 + l] = (float)s.cur[l]; out[20 + l] = s.swing_remaining[l];
    }
}

}  // namespace qro

extern "C" void qro_gait_run(const float *cfg19, int nticks, const float *time, const float *contact /*[nticks][4]*/, const int *stop, float *out /*[nticks][24]*/)
{
    qro::GaitConfig c;
    for (int l = 0; l < 4; ++l) { c.stance_duration[l] = cfg19[l]; c.duty_factor[l] = cfg19[4 + l]; c.initial_leg_phase[l] = cfg19[8 + l]; c.initial_leg_state[l] = (int)cfg19[12 + l]; }
    c.contact_detection_phase_threshold = cfg19[16]; c.wait_time = cfg19[17]; c.advanced_trot = cfg19[18] != 0.f;
    qro::GaitState s;
    qro::gait_reset(c, s);
    for (int k = 0; k < nticks; ++k) qro::gait_update(c, time[k], contact + 4 * k, stop ? stop[k] != 0 : false, s, out + 24 * k);
}
