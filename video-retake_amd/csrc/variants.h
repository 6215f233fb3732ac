// variants.h - the compile-time knobs of A/B builds, in one place.  Production builds (csrc/Makefile) define NONE of
// them: the values below are the shipped configuration.  tools/variants.sh rebuilds the library with -D overrides into
// retake/_lib/variants/ (selected at run time with RETAKE_HIP_LIB=...), which is how every same-box A/B in
// profiles/*_ab_*.txt was taken.
#pragma once

#ifndef RTK_P1_NB        // 32-row register blocks per wave, pass 1 of the bf16 LDS-DMA kernels (1 or 2)
#define RTK_P1_NB 2
#endif
#ifndef RTK_P1_LAZY      // RowStatB::update_lazy instead of the eager online max
#define RTK_P1_LAZY true
#endif
#ifndef RTK_P2_NB        // 32-key register blocks per wave, pass 2 (1 or 2)
#define RTK_P2_NB 2
#endif
#ifndef RTK_FORCE_KS     // > 0: key splits of pass 1 (bf16 path) instead of score_ws()'s shape rule
#define RTK_FORCE_KS 0
#endif
#ifndef RTK_FORCE_RS     // > 0: row splits of pass 2 (bf16 path) instead of score_ws()'s shape rule
#define RTK_FORCE_RS 0
#endif
#ifndef RTK_IGNORE_MANY_UNITS   // 1 = RTK_SCORE_MANY_UNITS changes nothing (A/B of the split policy of batched launches)
#define RTK_IGNORE_MANY_UNITS 0
#endif
#ifndef RTK_P1_RAW        // exact modes' pass 1: 1 = plain row sums + end-of-row check + fix-up launch, 0 = lazy online max (RowStatB)
#define RTK_P1_RAW 1
#endif
#ifndef RTK_PREP_NW       // fused prepare kernel, 16-bit dtypes: 32-bit words per thread and row half (1, 2 or 4)
#define RTK_PREP_NW 1
#endif
#ifndef RTK_PREP_HU       // fused prepare kernel: query heads whose rows are requested together (register batch)
#define RTK_PREP_HU 7
#endif
#ifndef RTK_PREP_YSPLIT   // fused prepare kernel: the query heads are split over this many workgroups per token range (>= 2: k and v go to the first and the last)
#define RTK_PREP_YSPLIT 2
#endif
#ifndef RTK_REF_P1_NB     // reference-rounding pass 1: 32-row register blocks per wave (1: 4 waves per SIMD, 2: 3)
#define RTK_REF_P1_NB 2
#endif
#ifndef RTK_REF_P2_NB     // reference-rounding pass 2: 32-key register blocks per wave
#define RTK_REF_P2_NB 1
#endif
#ifndef RTK_PREP_BLOCK    // threads per workgroup of the per-update kernels (fused prepare, attention prologue)
#define RTK_PREP_BLOCK 64
#endif
#ifndef RTK_PREP_UBASE    // per-update kernels: 1 = a row's address is (wave-uniform head base, kept in SGPRs) + (the thread's byte
#define RTK_PREP_UBASE 1  // offset, computed once); 0 = the full (h * stride_h + l * stride_l) * ES per head and access (A/B: profiles/r15_ab_prepare_addressing.txt)
#endif
#ifndef RTK_CMP_HU        // in-place compaction: KV heads per workgroup (register batch of the row loads)
#define RTK_CMP_HU 4
#endif
#ifndef RTK_CMP_WAVES     // in-place compaction: waves per SIMD the register allocation is bounded for
#define RTK_CMP_WAVES 4
#endif
#ifndef RTK_FIXUP_PROBE   // measurement only: 1 = the pass-1 fix-up launch returns at entry, 2 = it scans for NaNs but never repairs
#define RTK_FIXUP_PROBE 0
#endif
