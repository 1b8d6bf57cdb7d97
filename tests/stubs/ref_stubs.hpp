// Minimal stand-ins for the reference types the adapters touch (Eigen and ROS are not installed here).
// Only the members used by include/qrgpu_adapters.hpp exist.  TEST INFRASTRUCTURE.
#pragma once
#include <array>
#include <vector>

template <typename T> struct Vec3 { T v[3]; T &operator[](int i) { return v[i]; } const T &operator[](int i) const { return v[i]; } };
template <typename T> struct Quat { T v[4]; T &operator[](int i) { return v[i]; } const T &operator[](int i) const { return v[i]; } };   // (w,x,y,z), qr_cpptypes.h:83-84
template <typename T> struct Vec4b { bool v[4]; bool operator[](int i) const { return v[i]; } };
struct Mat34f { float m[12]; float *data() { return m; } float &operator()(int r, int c) { return m[3 * c + r]; } float operator()(int r, int c) const { return m[3 * c + r]; } };                  // column-major like Eigen
template <typename T> struct Vec12 { T v[12]; T &operator[](int i) { return v[i]; } const T &operator[](int i) const { return v[i]; } };

struct qrWbcCtrlData {          // quadruped/include/quadruped/controllers/qr_state_dataflow.h:133-192
    Vec3<float> pBody_des, vBody_des, aBody_des, pBody_RPY_des, vBody_Ori_des;
    Vec3<float> pFoot_des[4], vFoot_des[4], aFoot_des[4], Fr_des[4];
    Vec4b<bool> contact_state;
    bool allowAfterMPC = true;
};

struct qrRobotStub {            // the getters qrWbcLocomotionController::UpdateModel uses (:141-146)
    Quat<float> ori; Vec3<float> pos, vb, rpyrate; Vec12<float> q, dq;
    Quat<float> GetBaseOrientation() const { return ori; }
    Vec3<float> GetBasePosition() const { return pos; }
    Vec3<float> GetBaseVelocityInBaseFrame() const { return vb; }
    Vec3<float> GetBaseRollPitchYawRate() const { return rpyrate; }
    Vec12<float> GetMotorAngles() const { return q; }
    Vec12<float> GetMotorVelocities() const { return dq; }
    Mat34f footBase;
    Mat34f GetFootPositionsInBaseFrame() const { return footBase; }
};
