/* zkir_amd_experimental.h — measurement probes and kernel experiments exported by libzkir_amd.so that are NOT part of the drop-in boundary (include/zkir_amd.h):
 * nothing in the reference has a counterpart, no caller of the boundary needs them, and they may change or disappear.  Used by bench.py (the peaks) and scripts/ (the experiments). */
#ifndef ZKIR_AMD_EXPERIMENTAL_H
#define ZKIR_AMD_EXPERIMENTAL_H
#include "zkir_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* diagnostic: measured peak rate (per second) of independent Montgomery multiplications on the current device — the integer-ALU
 * roofline the Poseidon2 kernels are priced against (they are ALU-bound, not HBM- or MFMA-bound) */
double zkir_modmul_peak_per_s(void* hip_stream);
/* diagnostic: achieved HBM bandwidth (GB/s, read + write bytes) of a 16-byte-per-lane grid-stride device-to-device copy of `bytes` bytes (rounded down to 16 KiB; two scratch
 * buffers are allocated and freed): the measured copy peak the HBM-bound stages are held against next to the nominal 8 TB/s (MI355X_MICROARCH.md: ~6.3 TB/s for this copy) */
double zkir_hbm_copy_peak_gbs(uint64_t bytes, void* hip_stream);
/* EXPERIMENT (profiles/HISTORY.md, round 4): zkir_main_trace_launch + zkir_lde_launch (default VM mode) with the first two blocks of the main trace never written: the
 * extension's first inverse pass generates them from the trace.  m = scratch for the main-trace matrix (as zkir_main_trace_launch's out), out = the LDE.
 * Same output as the two calls; ZKIR_ERR_ARGUMENT where it does not apply (padded log2 rows < 20 or = 21).  Measured in profiles/r04*_fused01*. */
int zkir_commit_fused01_launch(const zkir_stark_ctx* ctx, const zkir_trace_columns* trace, uint64_t n_real, uint32_t* m, uint32_t width, uint32_t* out, void* hip_stream);
/* EXPERIMENT: one strided NTT pass (stage 0 of the inverse transform over 2^log_n rows, or stage 11 of the forward one over 2^(log_n + 1)) with tile geometry `variant`
 * (ntt.hip: strided_variant_run lists them) over `width` columns: for timing the tilings side by side (scripts/time_ntt_tiles.py); the data is left partially transformed. */
int zkir_ntt_strided_variant_launch(const zkir_stark_ctx* ctx, uint32_t* data, uint32_t width, int variant, int forward, void* hip_stream);

/* EXPERIMENT (round 6; profiles/r06_overlap_variants.txt): zkir_lde_launch + zkir_merkle_commit_launch of a canonical matrix with the leaf sponge absorbed block-group-wise on
 * the context's second stream while the next group of `group` B8 blocks is extended on `hip_stream` (a 12-word sponge state per leaf travels through HBM between groups).
 * in / out as zkir_lde_launch, tree as zkir_merkle_commit_launch; same output as the two calls. */
int zkir_commit_overlapped_launch(const zkir_stark_ctx* ctx, uint32_t* in, uint32_t width, uint32_t* out, uint32_t* tree, uint32_t group, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
