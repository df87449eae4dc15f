/* pyfem_hip_bench.h -- MEASUREMENT helpers of bench.py and tools/ (not part of the drop-in boundary of pyfem_hip.h;
 * nothing on the product path calls them).  They are built into libpyfem_hip.so from their own translation unit,
 * csrc/pf_bench.hip, because they have to launch the library's kernels on the library's stream with HIP events around
 * them.  No counterpart in the reference. */
#ifndef PYFEM_HIP_BENCH_H
#define PYFEM_HIP_BENCH_H
#include "pyfem_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- timing helper: average device time (ms) of `reps` back-to-back launches of one
 *      kernel family on the ctx stream, measured with HIP events (bench.py roofline leg). -- */
enum { PF_K_ASSEMBLE_ELASTICITY = 0, PF_K_SPMV = 1, PF_K_ASSEMBLE_HELMHOLTZ = 2 };
int pf_time_kernel(pf_ctx* ctx, int which, int reps, const double* in0, const double* in1,
                   double* out0, double E, double nu, double p, double* ms_host);

/* ---- HBM ceiling of THIS device in THIS process (bench.py: `frac_of_measured` next to the spec-sheet fraction).
 *      Streams `nbytes` of scratch (allocated and freed inside) `reps` times with persistent waves that each walk whole
 *      64 KiB runs with 16-byte lane accesses (the friendliest walk: scattered 1 KiB pieces reach a third less);
 *      ms_host = average milliseconds per pass (HIP events).  No counterpart in the reference.
 *      mode PF_HBM_FILL: write only;  PF_HBM_READ: read only;  PF_HBM_MIX: one quarter read, three quarters
 *      written (the byte mix of the quad assembly: K written once, records + coordinates read).
 *      nt: 0 plain accesses, 1 non-temporal loads and stores, 2 non-temporal stores only (what the assembly does). */
enum { PF_HBM_FILL = 0, PF_HBM_READ = 1, PF_HBM_MIX = 2 };
int pf_hbm_probe(pf_ctx* ctx, int mode, int nt, int64_t nbytes, int reps, double* ms_host);

#ifdef __cplusplus
}
#endif
#endif /* PYFEM_HIP_BENCH_H */
