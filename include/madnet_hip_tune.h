/* madnet_hip_tune.h -- process-wide tuning hooks of libmadnet_hip.so (atomics; meant for scripts/microbench.py and bench.py --set tune.*).
 * NOT part of the drop-in boundary (include/madnet_hip.h): nothing in the reference has a counterpart; a caller that only replaces the
 * reference's operators never needs this header. */
#ifndef MADNET_HIP_TUNE_H
#define MADNET_HIP_TUNE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int mh_tune_conv_tile(int bm, int bn);   /* tiled implicit-GEMM kernel: force the bm x bn tile (bits 0-15; 0 = heuristic), K-tile = bn >> 16; bm bits 16-19 switch off the
                                            uniform-tap loader / the intra-workgroup split-K / the stride-2 parity classes / the ragged-K uniform-tap instances (A/B) */
int mh_tune_conv_thin(int min_pixels);   /* weights-stationary thin-layer kernel from this many output pixels (0 = default, < 0 = never) */
int mh_tune_conv_patch(int mode);        /* patch-staged bf16 kernel of the stride-1 3x3 layers: 0 = off, 1 = on (tile heuristic), 64 / 128 = forced pixel tile, +256 = 8-wave variant, +2048 = generic-K instances only, +4096 = forward layers only, +8192 = no generic-K input gradients, < 0 = built-in default; returns the number of launches of that kernel since the previous call */
int mh_tune_conv_x3_igemm(int on);       /* split-bf16 (precision 2) on the tiled implicit-GEMM kernel for forward layers without a patch / bank instance: 0 = exact fp32 there (default: measured faster), 1 = on, < 0 = default */
int mh_tune_conv_rows(int min_pixels);   /* row-streaming kernel of the thin 3x3 stride-1 layers (conv_rows.hip: <= 16 input, <= 32 output channels): takes layers of at least this many output pixels (0 = never, < 0 = default 65536); returns the previous setting (-1 = default not resolved yet) */
int mh_tune_conv_bank_tile(int max_wgs);     /* split-bf16 bank kernel: layers whose 64x128 / 128x64 grid would have fewer workgroups than this take the 64x64 tile with 4 waves (0 = never, < 0 = default 200); returns the previous setting */
int mh_tune_conv_bank_small(int mode);   /* workgroup placement of the small-layer bank kernel: 0 = every XCD, pixel-major order (round 5); 1 (default, also < 0) = the logical order with the least L2 fill traffic (column-major where the bank outweighs the input), all XCDs; 2 = the layer confined to the fewest XCDs that give every workgroup its own CU, pixel-major; 3 = both.  Measured time-neutral (profiles/r06_microbench_small_placement.txt).  Returns the previous setting */
int mh_tune_conv_bank(int small_maxpix); /* fragment-bank kernels (mh_conv2d_wb): the small-layer kernel takes layers of up to this many output pixels (0 = never, < 0 = default 4096); returns the number of bank-kernel launches since the previous call */
int mh_tune_wgrad_wgs(int target_workgroups);
int mh_tune_wgrad_target_pct(int pct);   /* scale (percent) of the filter-gradient pixel-split workgroup targets for the split counts resolved from now on (a plan stores the counts it was recorded with); 0 = default.  Returns the PREVIOUS value (NOT a status code) so that a caller can scope the setting: DispNet's engine records with 150 under a process-wide lock and restores what it found */
int mh_tune_wgrad_image(int on);        /* image-layer filter-gradient kernel (3x3, Cin <= 3, Cout = 16, bf16; wgrad.hip): 0 = off (default: not yet timed on the GPU), 1 = on, > 1 = on with this many workgroups; returns the previous setting */
int mh_tune_wgrad_stream(int dist);     /* prefetch distance (row groups in flight) of the streaming filter-gradient kernel: 1 or 2, 0 = default */
int mh_tune_corr(int direct);            /* bit 0 (default SET): the direct small-D kernels; bit 1: plain instead of XCD-aware workgroup order of the large-D kernels; bit 2: the large-D bf16 gradient as two launches; bit 3: one fine row per workgroup in mh_level_front_head_fwd (default: two, when H = 2 Hc).  Default state = mh_tune_corr(1) */
int mh_tune_corr_row(int on);            /* backward front end of a pyramid level (mh_corr_warp_bwd): 1 = row-owned form, the warp-gradient scatter on an LDS copy of the row, the row's operands staged in LDS (default), 3 = row-owned without the operand staging, 0 = the global-atomic form */

int mh_tune_conv_planes(int mode);       /* pre-split-operand forward kernel (mh_conv2d_planes): bits 0-3 = tile variant (0 = heuristic), bit 4 / bit 5 = the staggered 128-pixel tile (two out-of-phase wave groups of one workgroup) for forward layers / input gradients (default: off), bit 7 = ... only where the tile has 128 columns, bit 8 = skip the K walk, bit 9 = skip the patch staging, bit 12 = skip the epilogue, bit 13 = epilogue without its stores, bit 14 = staggered groups without s_setprio, bit 15 = s_setprio 2 for the plain kernel (timing experiments: scripts/microbench.py phases); returns the number of launches of the plane kernels since the previous call */

#ifdef __cplusplus
}
#endif
#endif
