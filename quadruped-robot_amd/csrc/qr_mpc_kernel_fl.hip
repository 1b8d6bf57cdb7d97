// The MPC kernels once more, with the executed-arithmetic counters compiled in (qrgpu_enable_flop_count): see QR_FLOPS_BUILD in
// qr_mpc_kernel.hip.  Launched only while counting is on; the timed path runs the kernels of qr_mpc_kernel.hip.
#define QR_FLOPS_BUILD 1
#include "qr_mpc_kernel.hip"
