/* lsq_internal.h -- NOT part of the drop-in boundary (include/lsq_hip.h is), and NOT in the production library:
 * these symbols exist only in the tools build, tools/_tune/liblsq_hip_tools.so (`make tools`, -DLSQ_TOOLS), which the A/B
 * scripts under tools/ and the branch-pinning tests load through tools/lsq_tools.py.
 * `_ex` twins of the four ops with one extra argument, a launch-variant code
 * (unroll | nt << 8 | pipelined << 10 | loop form << 12 | blocks_per_cu << 16; 0 = the tuned default); without
 * -DLSQ_TUNING only the loop form and blocks_per_cu are honoured.
 * lsq_hip_debug_*: process-wide overrides of the launch policy (lsq_kernels.hpp `knob`). */
#ifndef LSQ_INTERNAL_H_
#define LSQ_INTERNAL_H_
#include "../../include/lsq_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
int lsq_hip_forward_per_tensor_ex(int dtype, const void* x, void* y, int64_t n, const void* scale, const void* shift,
                                  const lsq_params* p, const lsq_fwd_extras* extras, void* stream, int variant);
int lsq_hip_backward_per_tensor_ex(int dtype, const void* grad, const void* x, void* dx, void* ds, void* db,
                                   double* dsdb_wide, int64_t n, const void* scale, const void* shift,
                                   const lsq_params* p, const lsq_bwd_extras* extras, void* workspace,
                                   size_t workspace_bytes, void* stream, int variant);
int lsq_hip_forward_per_channel_ex(int dtype, const void* x, void* y, int64_t outer, int64_t channels, int64_t inner,
                                   const void* scale, const void* shift, const lsq_params* p,
                                   const lsq_fwd_extras* extras, void* stream, int variant);
int lsq_hip_backward_per_channel_ex(int dtype, const void* grad, const void* x, void* dx, void* ds, void* db,
                                    double* dsdb_wide, int64_t outer, int64_t channels, int64_t inner,
                                    const void* scale, const void* shift, const lsq_params* p,
                                    const lsq_bwd_extras* extras, void* workspace, size_t workspace_bytes, void* stream,
                                    int variant);
/* [grid x, grid y, resident workgroups per CU used for the geometry, numRegs, kind (1 = 256-lane windows, 2 = row-group
 * windows, 3 = segment mode), LDS-DMA ring depth (0 = register loops), workgroup size, ring copies with the streaming hint]
 * of this thread's last per-channel backward launch */
void lsq_hip_debug_last_launch(int* out8);
/* tests / tools only: loop form of the window-mode per-channel kernels for calls that do not choose one (variant bits 12-13
 * zero): 0 = the built-in policy, 1 = register loops, 2 = LDS-DMA ring whatever the shape */
void lsq_hip_debug_force_ring(int v);
/* tools only: rows a row-group-window workgroup walks at least (0 = the built-in rule) */
void lsq_hip_debug_set_ww_min_rows(int v);
/* tools only: row-group windows cut rows of 128 / 192 / 256 lanes into 64-lane windows (4 row groups each) */
void lsq_hip_debug_set_ww_split64(int v);   /* 0 = default (on for 4/8-byte storage), 1 = on for all, 2 = off */
void lsq_hip_debug_set_ring_nt(int v);      /* streaming hint on the LDS-DMA copies: 0 = policy, 1 = on, 2 = off */
void lsq_hip_debug_set_ww_big(int v);       /* 1024-lane row-group workgroups: 0 = policy, 1 = always, 2 = never */
/* tools only: channels (slots) per finalize workgroup, a power of two <= 32 (0 = the built-in rule) */
void lsq_hip_debug_set_ww_max_log2(int v);  /* row-group windows for last-axis tensors below 2^v elements: 0 = policy */
void lsq_hip_debug_set_seg_min_div(int v);  /* forward / backward: segment mode for channel rows of at least 1/v of a workgroup's span (1 = whole spans only): 0 = policy */
void lsq_hip_debug_set_fwd_direct(int v);   /* last-axis forward: 0 / 3 = policy (lanes read their own scale / shift), 1 = the same on the usual grid, 2 = LDS table */
void lsq_hip_debug_set_seg_no_up_front(int v); /* segment kernels: 1 = always the loop form, never the one-group short walk */
void lsq_hip_debug_set_fin_ch(int v);
/* owner windows of the per-channel backward: 0 = the policy, 1 = wherever the shape allows, 2 = never,
   3 = like 1 but without the waves' turns at the higher issue priority (A/B) */
void lsq_hip_debug_set_own(int v);
/* owner windows: shortest run (bytes of one row an owner reads) the plan accepts; 0 = the default (kOwnMinRunBytes) */
void lsq_hip_debug_set_own_min_run(int v);
/* owner windows: channel group of the plan -- 0 = the policy, 1 = the smallest that works, 2 = the largest that still gives every CU an owner */
void lsq_hip_debug_set_own_fat(int v);
/* tuning only: override the workgroups-per-CU of the observer-statistics kernels (0 = defaults) */
void lsq_hip_debug_set_observe_wg_per_cu(int v);
#ifdef __cplusplus
}
#endif
#endif
