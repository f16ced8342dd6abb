/*
 * emspec_debug.h — diagnostic entry points of libemspec.  NOT part of the drop-in
 * boundary: nothing on the product path calls them.  They exist so that tests and
 * tools can (a) compare the kernels' two row-lookup implementations on arbitrary
 * inputs and (b) read per-phase cycle counts from the stamped (diagnostic) build of
 * the fused kernel.  [BUILD-DEFINED]; the reference has no counterpart.
 */
#ifndef EMSPEC_DEBUG_H
#define EMSPEC_DEBUG_H
#include "emspec.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Evaluate the hinted row lookup (fused/generic kernels) and the binary search on `count`
 * host values of k-hat for fft size n: out_hint[i], out_exact[i] = row or -1. */
int emspec_debug_row_lookup(emspec_engine* e, int32_t n, const float* kh, int64_t count,
                            int32_t* out_hint, int32_t* out_exact);

/* Evaluate the kernels' 7-instruction reciprocal (recip_normal, emspec_device.h) and the IEEE division 1.0f / d on
 * `count` host values: out_short[i], out_ieee[i].  They must agree bit for bit for 2^-96 < d < 2^126. */
int emspec_debug_recip(emspec_engine* e, const float* d, int64_t count, float* out_short, float* out_ieee);

/* The same for the EXACT mode's binary64 reciprocal (recip_normal64, exact_fused.hip.inc) and 1.0 / d: bit for bit equal for
 * 2^-700 < d < 2^1000. */
int emspec_debug_recip64(emspec_engine* e, const double* d, int64_t count, double* out_short, double* out_ieee);

/* Run the stamped build of the fused kernel on device-resident pcm and return, per
 * workgroup and wave, the shader-clock cycles spent in each barrier-delimited phase:
 * cycles[groups][waves][8] (host).  Call with cycles == NULL to get *groups / *waves.
 * On an EXACT-mode engine: the stamped build of exact_fused4096_kernel (slots: tools/phase_cycles_exact.py). */
int emspec_debug_phase_cycles(emspec_engine* e, const float* pcm_dev, int32_t S, int64_t L,
                              int32_t n, int32_t hop, int32_t reassign, float* db_dev,
                              uint8_t* index_dev, uint64_t* cycles, int64_t* groups,
                              int32_t* waves);

/* Synchronises the device and returns non-zero if a bounded spin of the decoupled-team fused
 * kernel (EMSPEC_FUSED_VARIANT=r8t) ever timed out: a protocol bug, results are then invalid. */
int emspec_debug_fused_error(emspec_engine* e);

/* Enqueue on hip_stream a kernel that keeps `groups` 256-thread workgroups resident for about
 * `usec` microseconds (bounded spin): a stand-in for a communication kernel holding compute
 * units while the column kernels run on another stream (tools/occupancy_probe.py). */
int emspec_debug_occupy(emspec_engine* e, int32_t groups, int32_t usec, void* hip_stream);

/* Live multi-stream calls: have the frame kernel's first workgroup of every stream stamp the 100 MHz wall clock at its
 * phases into stamps[streams][8] (page-locked host memory of the caller; NULL = off): 0 entry, 1 descriptor read, 2 samples
 * in registers, 3 transform done, 4 bins + scatter issued, 5 ticket taken, 6 finalised (tools/live_phases.py). */
int emspec_debug_live_stamps(emspec_engine* e, uint64_t* stamps);

#ifdef __cplusplus
}
#endif
#endif
