// jpeg2png_amd — declarations shared by the translation units of libjpeg2png_amd.so; not part of the C-ABI.
#pragma once
#include <stdlib.h>
#include "jpeg2png_amd.h"

// Environment knobs of the EXPERIMENTS build (-DJ2P_EXPERIMENTS: jpeg2png_amd/libjpeg2png_amd_exp.so, built by
// buildlib.build_experiments() for the schedule-equivalence tests and the timing tools): schedules that measured slower
// everywhere (one column per lane, all channels of a joint image in one wavefront, the reduction as the gradient
// launch's last workgroup, split phases) and the switches that select them.  The release library carries neither the
// kernels nor the switches and reads only J2P_DEVICE, J2P_DEVICES, J2P_TILED_EXCHANGE, J2P_TILED_WAIT,
// J2P_TILED_VERIFY, J2P_RCCL_LIBRARY, J2P_POOL_MIB and J2P_COMPUTE_TIMING.
static inline const char *j2p_exp_env(const char *name)
{
#ifdef J2P_EXPERIMENTS
        return getenv(name);
#else
        (void)name;
        return nullptr;
#endif
}

// error text of the calling thread (what j2p_last_error() returns); returns `code`
int j2p_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// log rows from per-iteration sums like j2p_log_rows_from_sums(), but continuing a run: carried[] holds the prob
// distance per channel of the state entering the first of the n iterations and is updated (all 0 at iteration 0);
// !carried_valid: that distance is unknown (the previous iterations ran without logging) and the first row
// reports NaN for prob_dist and objective, as j2p_solver_run does
void j2p_rows_from_sums_carry(unsigned nch, float weight, const float *pweight, unsigned n, const double *sums,
                              double *carried, bool carried_valid, j2p_log_row *rows);


// test hook behind j2p_debug_fail_run_after(): true when THIS run call is the one that has to fail
bool j2p_injected_failure();
// ... (negative argument) true when the LAST band of this threaded run has to fail halfway through its iterations
bool j2p_injected_band_failure();
