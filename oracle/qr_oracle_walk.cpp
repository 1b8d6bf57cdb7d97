// TEST INFRASTRUCTURE -- CPU restatement of the reference's walk gait generator and of the force-window ratios its sub-states select; only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
//   qrWalkGaitGenerator (constructor from YAML, Update)          QS/gait/qr_walk_gait_generator.cpp:66-193, 202-288
//   qrGaitGenerator::Reset                                      QI/gait/qr_gait.h:76-87
//   TorqueStanceLegController::UpdateFRatio, walk branch        QS/controllers/balance_controller/qr_torque_stance_leg_controller.cpp:125-168
//
// Plain float arithmetic with the reference's float / double mix.  Parity unpinned against a compiled reference (the class needs Eigen and
// yaml-cpp); pinned by the closed form of the schedule (tests/test_oracle_walk.py).  Members the reference leaves uninitialised until
// their first write (moveBasePhase, detectedLegState, detectedEventTickPhase) start at zero here.
#include "qr_oracle.h"

#include <cmath>

namespace qro {

enum { W_SWING = 0, W_STANCE = 1, W_EARLY = 2, W_LOSE = 3, W_LOAD = 5, W_UNLOAD = 6, W_FULL = 7, W_TRUE_SWING = 8 };

void walk_derive(const WalkConfig &c, WalkDerived &d)
{
    // :87-124: entries with a ratio below 0.01 are dropped; the stance-like sub-states in front of true_swing add up to its start
    d.nq = 0;
    float standRatioInSwing = 0.f;
    d.true_swing_start_in_swing = 0.f;
    for (int i = 0; i < c.n_states; ++i) {
        if (c.state_ratio[i] < 0.01) continue;
        if (c.state_switch[i] == W_TRUE_SWING) d.true_swing_start_in_swing = standRatioInSwing;
        else standRatioInSwing += c.state_ratio[i];
        d.que[d.nq] = c.state_switch[i]; d.ratio[d.nq] = c.state_ratio[i]; ++d.nq;
    }
    d.accum[0] = 0.f;
    for (int i = 0; i < d.nq; ++i) d.accum[i + 1] = d.accum[i] + d.ratio[i];
    for (int l = 0; l < 4; ++l) {
        d.full[l] = c.stance_duration[l] / c.duty_factor[l];
        d.state_index0[l] = 0;
        if (c.initial_leg_state[l] == W_SWING) {           // :146-157 (the division by dutyFactor is the reference's)
            const float ph = (c.initial_leg_phase[l] - c.duty_factor[l]) / c.duty_factor[l];
            int i = 0;
            while (i < d.nq && ph > d.accum[i]) i++;
            d.state_index0[l] = i - 1 > 0 ? i - 1 : 0;
        }
    }
}

void walk_reset(const WalkConfig &c, const WalkDerived &d, WalkState &s, bool constructed)
{
    for (int l = 0; l < 4; ++l) {
        s.nphase[l] = 0.f;
        s.cur[l] = s.leg[l] = s.desired[l] = c.initial_leg_state[l];
        if (constructed) { s.state_index[l] = d.state_index0[l]; s.phase[l] = 0.f; s.detected[l] = 0; s.event_phase[l] = 0.f; }
    }
    if (constructed) s.move_base_phase = 0.f;
}

// out[41]: phaseInFullCycle[4], normalizedPhase[4], desiredLegState[4], legState[4], curLegState[4], detectedLegState[4],
// detectedEventTickPhase[4], moveBasePhase, contacts[4], fMinRatio[4], fMaxRatio[4]
void walk_update(const WalkConfig &c, const WalkDerived &d, float currentTime, const float contact[4], bool stop, WalkState &s, float out[41])
{
    for (int l = 0; l < 4; ++l) {
        if (!stop || (stop && s.cur[l] == W_SWING)) s.cur[l] = s.desired[l];
        const float augmentedTime = c.initial_leg_phase[l] * d.full[l] + currentTime;
        s.phase[l] = std::fmod(augmentedTime, d.full[l]) / d.full[l];
        const float ratio = c.duty_factor[l];
        if (s.phase[l] <= c.duty_factor[l]) {
            if (s.cur[l] != W_STANCE) s.state_index[l] = 0;
            s.desired[l] = W_STANCE; s.leg[l] = W_STANCE;
            s.nphase[l] = s.phase[l] / ratio;
        } else {
            s.desired[l] = W_SWING; s.leg[l] = W_SWING;
            s.nphase[l] = (float)((s.phase[l] - ratio) / (1.0 - ratio));
        }
        if (s.desired[l] == W_SWING) {
            int idx = s.state_index[l];
            const float start = d.accum[idx], end = d.accum[idx + 1];
            const float psc = (float)((s.phase[l] - c.duty_factor[l]) / (1.0 - c.duty_factor[l]));
            if (psc <= end && psc >= start) {
                s.desired[l] = d.que[idx];
                s.nphase[l] = (psc - start) / (end - start);
            } else {
                idx += 1;
                if (idx > d.nq - 1) idx = d.nq - 1;          // (the reference indexes past its queue here; cannot happen while a tick is shorter than a sub-state)
                s.desired[l] = d.que[idx];
                s.state_index[l] = idx;
                s.nphase[l] = (psc - d.accum[idx]) / d.ratio[idx];
            }
            if (psc < d.true_swing_start_in_swing) s.move_base_phase = psc / d.true_swing_start_in_swing;
            else s.move_base_phase = 1.0f;
        }
        s.detected[l] = (s.desired[l] != W_STANCE) ? W_SWING : W_STANCE;
        if (s.nphase[l] < c.contact_detection_phase_threshold) continue;
        if (s.desired[l] == W_TRUE_SWING && contact[l] != 0.f) { s.detected[l] = W_EARLY; s.event_phase[l] = s.phase[l]; }
        else if (s.desired[l] == W_STANCE && contact[l] == 0.f) { s.detected[l] = W_LOSE; s.event_phase[l] = s.phase[l]; }
    }
    for (int l = 0; l < 4; ++l) {
        out[l] = s.phase[l]; out[4 + l] = s.nphase[l]; out[8 + l] = (float)s.desired[l]; out[12 + l] = (float)s.leg[l]; out[16 + l] = (float)s.cur[l];
        out[20 + l] = (float)s.detected[l]; out[24 + l] = s.event_phase[l];
        // UpdateFRatio, walk branch
        float phase = s.nphase[l], cont, fmax, fmin = 0.001f;
        if (s.detected[l] == W_STANCE || s.detected[l] == W_LOSE) { cont = 1.f; fmax = 10.0f; }
        else if (s.detected[l] == W_EARLY) { cont = 1.f; const float t = std::abs(phase - 0.8f); fmax = 10.0f * std::min(0.01f, t); }
        else if (s.desired[l] == W_LOAD) { cont = 1.f; fmax = 10.0f * std::max(0.001f, phase); }
        else if (s.desired[l] == W_UNLOAD) { cont = 1.f; phase = phase / (3.f / 4.0f); fmax = 10.0f * std::max(0.001f, 1.0f - phase); }
        else if (s.desired[l] == W_TRUE_SWING) { cont = 0.f; fmax = 0.002f; }
        else { cont = 1.f; fmax = 10.0f; }                    // FULL_STANCE
        out[29 + l] = cont; out[33 + l] = fmin; out[37 + l] = fmax;
    }
    out[28] = s.move_base_phase;
}

}  // namespace qro

// cfg27: stance_duration[4], duty_factor[4], initial_leg_phase[4], initial_leg_state[4], contact_detection_phase_threshold, n_states,
//        state_switch[4] (SubLegState values), state_ratio[4], pad
extern "C" void qro_walk_run(const float *cfg, int nticks, const float *time, const float *contact /*[nticks][4]*/, const int *stop, float *out /*[nticks][41]*/)
{
    qro::WalkConfig c;
    for (int l = 0; l < 4; ++l) { c.stance_duration[l] = cfg[l]; c.duty_factor[l] = cfg[4 + l]; c.initial_leg_phase[l] = cfg[8 + l]; c.initial_leg_state[l] = (int)cfg[12 + l]; }
    c.contact_detection_phase_threshold = cfg[16]; c.n_states = (int)cfg[17];
    for (int i = 0; i < 4; ++i) { c.state_switch[i] = (int)cfg[18 + i]; c.state_ratio[i] = cfg[22 + i]; }
    qro::WalkDerived d;
    qro::walk_derive(c, d);
    qro::WalkState s;
    qro::walk_reset(c, d, s, true);
    for (int k = 0; k < nticks; ++k) qro::walk_update(c, d, time[k], contact + 4 * k, stop ? stop[k] != 0 : false, s, out + 41 * k);
}
