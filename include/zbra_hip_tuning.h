/*
 * zbra_hip_tuning.h — TEST AND TUNING HOOKS of libzbra_hip.so.  Not part of the product ABI
 * (include/zbra_hip.h): nothing here has a counterpart in the reference, no binding of the
 * library should call it, and any of it may change or go.  The hooks change the launch policy of
 * the whole process, so that tests reach code paths that otherwise need hundreds of gigabytes
 * (the cut of a launch at the work-items of one dispatch) and so that kernel variants can be timed
 * interleaved in one process.  Each call publishes a new policy snapshot under a mutex; a library
 * call that is already running on another thread keeps the snapshot it started with.
 * All return 0 on success.
 */
#ifndef ZBRA_HIP_TUNING_H
#define ZBRA_HIP_TUNING_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Work-items one dispatch may have (0 = the hardware's 2^32 - 1).  Decode launches are cut and the
 * pack launch's grid is capped by it. */
int zbra_tuning_set_launch_limit(uint64_t work_items);

/* The warm-up route of narrow decode submissions: packed bytes per warmed chunk (0 = route off),
 * widest array in encoded bits per value that takes it, smallest submission in encoded bytes that
 * does.  Defaults 64 MiB, 6 bits, 8 MiB. */
int zbra_tuning_set_decode_warm(uint64_t chunk_bytes, uint32_t max_bits, uint64_t min_bytes);

/* Decode tile kernel shape: values per tile, threads per workgroup, nt (bit 0: nontemporal loads,
 * bit 1: nontemporal stores), direct (0 = LDS image filled through registers, 1 = LDS-free reads, 2 = two
 * tiles per workgroup whose packed bytes are brought in by LDS-DMA; 2048 x 512 only), xcd (tile order: 0 plain, 1 chunked, 2 chosen
 * by the submission's bits per value, >= 4 groups of that many tiles).  Returns 1 for a shape the
 * library was not built with. */
int zbra_tuning_set_decode_variant(int tile, int threads, int nt, int direct, int xcd);

/* Order of the Enum row split's write pass: -1 by size (default), 0 front to back, 1 back to front. */
int zbra_tuning_set_enum_reverse(int reverse);

#ifdef __cplusplus
}
#endif
#endif /* ZBRA_HIP_TUNING_H */
