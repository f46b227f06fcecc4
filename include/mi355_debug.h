/* mi355_debug.h - diagnostic entry points of libmi355dbg.so (tools/ only; NOT part of the product ABI of
 * include/mi355_det.h and never loaded by the package, bench.py or __graft_entry__.smoke()). */
#ifndef MI355_DEBUG_H
#define MI355_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef void* mi_dbg_stream_t; /* hipStream_t */
/* a kernel of `kb` KiB (8 / 16 / 32 / 64 / 128) of straight-line scalar no-ops on `blocks` single-wave blocks - displaces
 * that much of every instruction cache without touching data (tools/icache_probe.py, tools/ctx_probe.py) */
int mi_debug_code_polluter(int kb, int blocks, mi_dbg_stream_t s);
/* where the blocks of a launch on stream s run: `blocks` blocks of 256 threads that each spin `spin_us` microseconds (so
 * that the whole grid is resident at once) and write out[2 * block] = XCC id, out[2 * block + 1] = HW_ID register (CU / SH /
 * SE fields) - the census that shows what a hipExtStreamCreateWithCUMask mask selects (tools/cu_mask_probe.py) */
int mi_debug_cu_census(uint32_t* out, int blocks, int spin_us, mi_dbg_stream_t s);
#ifdef __cplusplus
}
#endif
#endif
