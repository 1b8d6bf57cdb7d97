// Drives the header-only adapters exactly the way MPCStanceLegController::SolveDenseMPC and
// qrWbcLocomotionController::Run do, on inputs read from stdin; prints forces and torques.
// TEST INFRASTRUCTURE (tests/test_adapters.py).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ref_stubs.hpp"
#include "qrgpu_adapters.hpp"

// `adapter_demo --latency N`: after the one checked call of each adapter, N more timed calls of SolveMPCKernel (+ the twelve GetMPCSolution
// reads SolveDenseMPC makes) and of WbcRun on the same inputs; prints "latency_mpc_us p50 p99 mean min" and "latency_wbc_us ..." (bench.py --mode single).
static void print_latency(const char *name, std::vector<double> &us)
{
    std::sort(us.begin(), us.end());
    double mean = 0; for (double x : us) mean += x; mean /= (double)us.size();
    printf("%s %.2f %.2f %.2f %.2f\n", name, us[us.size() / 2], us[(us.size() * 99) / 100], mean, us.front());
}

int main(int argc, char **argv)
{
    const int lat_n = (argc >= 3 && !strcmp(argv[1], "--latency")) ? atoi(argv[2]) : 0;
    int h;
    float cfg[20], s[28], traj[12 * 16], gait[4 * 16], fb[37], cmd[67];
    if (scanf("%d", &h) != 1) return 2;
    for (float &x : cfg) if (scanf("%f", &x) != 1) return 2;
    for (float &x : s) if (scanf("%f", &x) != 1) return 2;
    for (int i = 0; i < 12 * h; ++i) if (scanf("%f", &traj[i]) != 1) return 2;
    for (int i = 0; i < 4 * h; ++i) if (scanf("%f", &gait[i]) != 1) return 2;
    for (float &x : fb) if (scanf("%f", &x) != 1) return 2;
    for (float &x : cmd) if (scanf("%f", &x) != 1) return 2;

    printf("before %g\n", Quadruped::GetMPCSolution(0));
    Quadruped::SetupProblem(cfg[0], h, cfg[1], cfg[2], cfg[3], cfg + 4, cfg + 7, cfg[19]);       // Reset(), :90
    Vec3<float> p{{s[0], s[1], s[2]}}, v{{s[3], s[4], s[5]}}, w{{s[10], s[11], s[12]}}, rpy{{s[25], s[26], s[27]}};
    Quat<float> q{{s[6], s[7], s[8], s[9]}};
    Mat34f r;
    for (int i = 0; i < 12; ++i) r.m[i] = s[13 + i];
    Quadruped::SolveMPCKernel(p, v, q, w, r, rpy, traj, gait);                                      // SolveDenseMPC, :399
    printf("force");
    for (int leg = 0; leg < 4; ++leg) for (int ax = 0; ax < 3; ++ax) printf(" %.9g", Quadruped::GetMPCSolution(leg * 3 + ax));   // :404
    printf("\n");
    if (lat_n > 0) {
        std::vector<double> us;
        double sink = 0;
        for (int it = 0; it < lat_n + 20; ++it) {
            const auto t0 = std::chrono::steady_clock::now();
            Quadruped::SolveMPCKernel(p, v, q, w, r, rpy, traj, gait);
            for (int k = 0; k < 12; ++k) sink += Quadruped::GetMPCSolution(k);
            const auto t1 = std::chrono::steady_clock::now();
            if (it >= 20) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        print_latency("latency_mpc_us", us);
        if (sink == 12345.678) printf("\n");
    }

    if (qrgpu_adapters::WbcSetup(0.08505f, 0.2f, 0.2f) != 0) return 3;
    qrRobotStub robot;
    for (int i = 0; i < 4; ++i) robot.ori[i] = fb[i];
    for (int i = 0; i < 3; ++i) { robot.pos[i] = fb[4 + i]; robot.rpyrate[i] = fb[7 + i]; robot.vb[i] = fb[10 + i]; }
    for (int i = 0; i < 12; ++i) { robot.q[i] = fb[13 + i]; robot.dq[i] = fb[25 + i]; }
    qrWbcCtrlData d;
    for (int i = 0; i < 3; ++i) { d.pBody_des[i] = cmd[i]; d.vBody_des[i] = cmd[3 + i]; d.aBody_des[i] = cmd[6 + i]; d.pBody_RPY_des[i] = cmd[9 + i]; d.vBody_Ori_des[i] = cmd[12 + i]; }
    for (int l = 0; l < 4; ++l) {
        for (int i = 0; i < 3; ++i) {
            d.pFoot_des[l][i] = cmd[15 + 3 * l + i]; d.vFoot_des[l][i] = cmd[27 + 3 * l + i]; d.aFoot_des[l][i] = cmd[39 + 3 * l + i];
            d.Fr_des[l][i] = (float)Quadruped::GetMPCSolution(3 * l + i);                            // wbcData.Fr_des[leg] = f.col(leg), :408
        }
        d.contact_state.v[l] = cmd[63 + l] != 0.f;
    }
    Vec12<float> tau, qd, qdd;
    int st = qrgpu_adapters::WbcRun(&robot, &d, tau, qd, qdd);
    printf("status %d\ntau", st);
    for (int i = 0; i < 12; ++i) printf(" %.9g", tau[i]);
    printf("\n");
    if (lat_n > 0) {
        std::vector<double> us;
        for (int it = 0; it < lat_n + 20; ++it) {
            const auto t0 = std::chrono::steady_clock::now();
            qrgpu_adapters::WbcRun(&robot, &d, tau, qd, qdd);
            const auto t1 = std::chrono::steady_clock::now();
            if (it >= 20) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        print_latency("latency_wbc_us", us);
    }

    // force-balance controller: TorqueStanceLegController::GetAction -> ComputeContactForce (qr_torque_stance_leg_controller.cpp:500)
    float vin[37];
    bool have_vmc = true;
    for (float &x : vin) if (scanf("%f", &x) != 1) { have_vmc = false; break; }
    if (have_vmc) {
        qrgpu_vmc_desc vd; qrgpu_vmc_desc_default(&vd);
        if (qrgpu_adapters::VmcSetup(vd) != 0) return 4;
        for (int i = 0; i < 12; ++i) robot.footBase.m[i] = vin[i];
        float acc[6]; bool ct[4];
        for (int i = 0; i < 6; ++i) acc[i] = vin[12 + i];
        for (int i = 0; i < 4; ++i) ct[i] = vin[18 + i] != 0.f;
        Mat34f F;
        int vst = qrgpu_adapters::VmcContactForce(&robot, vin + 22, vin + 31, vin + 34, acc, ct, F);
        printf("vmcstatus %d\nvmcforce", vst);
        for (int i = 0; i < 12; ++i) printf(" %.9g", F.m[i]);
        printf("\n");
        // world-frame overload: 8 more numbers (fMinRatio[4], fMaxRatio[4]); Rcb of the input is then rotMat
        float ratio[8];
        bool have_world = true;
        for (float &x : ratio) if (scanf("%f", &x) != 1) { have_world = false; break; }
        if (have_world) {
            Mat34f FW;
            int wst = qrgpu_adapters::VmcContactForceWorld(&robot, vin + 22, acc, ct, ratio, ratio + 4, FW);
            printf("vmcwstatus %d\nvmcwforce", wst);
            for (int i = 0; i < 12; ++i) printf(" %.9g", FW.m[i]);
            printf("\n");
        }
    }
    return 0;
}
