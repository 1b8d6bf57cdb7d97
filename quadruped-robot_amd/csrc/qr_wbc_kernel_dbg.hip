// The WBC kernel once more with the inspection outputs (qrgpu_wbc_dynamics_batch) and the cycle stamps (qrgpu_debug_cycles) compiled in:
// see QR_WBC_DBG_BUILD in qr_wbc_kernel.hip.  Launched only when a call asks for either.
#define QR_WBC_DBG_BUILD 1
#include "qr_wbc_kernel.hip"
